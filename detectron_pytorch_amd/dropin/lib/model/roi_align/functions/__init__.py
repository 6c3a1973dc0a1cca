"""Overlay package: merged with the reference's package of the same name when both `dropin/lib` and the reference's
`lib/` are on sys.path (this directory first), so that only the modules present here are replaced."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

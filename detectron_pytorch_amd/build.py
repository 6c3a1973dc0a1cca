"""Build libmi_detectron_ops.so (the C-ABI of include/mi_detectron_ops.h) for gfx950 with hipcc.

Replaces the reference's lib/make.sh:5-63 (nvcc -gencode sm_30..sm_61 per op, then one
torch.utils.ffi.create_extension per op): a single `hipcc --offload-arch=gfx950 -shared` of the
.hip translation units in csrc/, no torch headers, no cffi.  hipcc cross-compiles without a GPU,
so this runs in the CPU-only build container; the .so is git-ignored and travels to the GPU box
inside the gpurun snapshot.

Flags that matter:
  -ffp-contract=off        fp32 results must match the CPU oracle operation for operation
                           (NMS decisions at IoU == thresh)
  -munsafe-fp-atomics      fp32 atomicAdd lowers to the hardware global_atomic_add_f32 instead of
                           a compare-and-swap loop (backward scatter kernels)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libmi_detectron_ops.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["abi.hip", "roi_align.hip", "roi_align_records.hip", "roi_align_nhwc.hip", "roi_pool.hip", "roi_crop.hip", "nms.hip", "soft_nms.hip", "proposals.hip", "affine_channel.hip", "topk.hip", "results.hip", "box_voting.hip", "mask_targets.hip"]
ARCH = "gfx950"
# Per-unit flags.  roi_align_records.hip: the leading scalar / pointer kernel arguments arrive preloaded in SGPRs (gfx950
# kernarg preload) -- the records-free RoIAlign forward starts with a chain of dependent fetches (arguments -> the RoI's five
# floats -> geometry -> window), and this takes the first link out of it.
UNIT_FLAGS = {"roi_align_records.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
              "roi_align_nhwc.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def flags():
    out = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
           "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if os.environ.get("MI_EXTRA_DEFINES"):  # tools/ only: compile-time variants of one kernel for an A/B arm
        out += ["-D" + d for d in os.environ["MI_EXTRA_DEFINES"].split()]
    if os.environ.get("MI_TUNING_BUILD"):  # tools/ only: keeps the MI_ROI_ALIGN_ABLATE switches alive in the kernels
        out.append("-DMI_TUNING=1")
    return out


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(ROOT, "include", "mi_detectron_ops.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip unit to an object (in parallel) and link the shared library."""
    if not force and not stale():
        return LIB_PATH
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + flags() + UNIT_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
        objs.append(obj)
    tmp = LIB_PATH + ".tmp"
    subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", tmp])
    os.replace(tmp, LIB_PATH)
    if verbose:
        print("built", os.path.relpath(LIB_PATH, ROOT))
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)

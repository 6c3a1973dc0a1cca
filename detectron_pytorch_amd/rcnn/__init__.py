"""The consumer graph of the hot path: Generalized R-CNN (ResNet-FPN backbone, RPN, box / mask / keypoint heads, losses)
on stock PyTorch-ROCm convolutions, wired to the HIP operators of this package (SURVEY.md section 2a rows 7-9, 8d
configs 3-5).  Module and parameter names are the reference's, so a reference `state_dict` loads unchanged.
"""

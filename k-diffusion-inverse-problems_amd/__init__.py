"""kdip_amd -- MI355X-native guided-diffusion inverse-problem sampler hot path.

Host-side mirror of the reference's plug-in surfaces over the C-ABI library
`libkdip_hip.so` (hand-written HIP for gfx950):

    kdip_amd.sampling      <- k_diffusion/sampling.py      (get_sigmas_karras, sample_euler, sample_heun)
    kdip_amd.external      <- k_diffusion/external.py      (DiscreteSchedule sigma<->t, OpenAIDenoiser[V2])
    kdip_amd.condition     <- condition/condition.py       (ConditionDenoiser*, register_mat_solver)
    kdip_amd.measurements  <- condition/measurements.py    (register_operator, get_operator, operators)
    kdip_amd.transforms    <- condition/utils.py           (OrthoTransform)
    kdip_amd.unet          <- guided_diffusion/{unet,script_util,gaussian_diffusion}.py (model + diffusion tables)
    kdip_amd.evaluation    <- k_diffusion/evaluation.py    (compute_features: shard + all_gather)

There is no CPU fallback: importing any compute module without the built library raises.
"""
__version__ = "0.1.0"

"""Measurement scripts take their knobs from the environment; the LIBRARY does not read it (include/ecgpu.h: ecgpu_set_option),
so the scripts translate:  ECGPU_MSM_CBITS, ECGPU_MSM_ROUNDS, ECGPU_MSM_SMALL, ECGPU_MSM_SLAB, ECGPU_FB_WINDOW,
ECGPU_FB_MAX_WINDOW, ECGPU_K256_FAST_WAVES -> per-context options."""
import os


def apply_env_options(ctx):
    import ecgpu
    for env, opt in (("ECGPU_MSM_CBITS", ecgpu.OPT_MSM_WINDOW_BITS), ("ECGPU_MSM_ROUNDS", ecgpu.OPT_MSM_ROUNDS),
                     ("ECGPU_MSM_SMALL", ecgpu.OPT_MSM_SMALL_PATH), ("ECGPU_MSM_SLAB", ecgpu.OPT_MSM_SLAB_TERMS),
                     ("ECGPU_FB_WINDOW", ecgpu.OPT_FB_WINDOW), ("ECGPU_FB_MAX_WINDOW", ecgpu.OPT_FB_MAX_WINDOW),
                     ("ECGPU_K256_FAST_WAVES", ecgpu.OPT_K256_WAVES)):
        v = os.environ.get(env)
        if v is not None and v != "":
            ctx.set_option(opt, int(v))

"""VO configuration: an attribute-dict with the keys of the reference's yacs node
(reference: ramp/config.py:3-27; yaml files in config_vo/)."""
import copy

import yaml

_DEFAULTS = dict(
    BUFFER_SIZE=2048, GRADIENT_BIAS=True, PATCHES_PER_FRAME=80, REMOVAL_WINDOW=20,
    OPTIMIZATION_WINDOW=12, PATCH_LIFETIME=12, KEYFRAME_INDEX=4, KEYFRAME_THRESH=12.5,
    MOTION_MODEL='DAMPED_LINEAR', MOTION_DAMPING=0.5, MIXED_PRECISION=True,
    # not a reference key (BASELINE configs[4]): with MIXED_PRECISION, the conv towers' products run on the fp8 MFMA
    # (fp16 storage, e4m3 operands, fp32 accumulate); off by default
    ENCODER_FP8=False)

# the reference's shipped presets (config_vo/{default,precise,fast}.yaml)
PRESETS = {
    "default": dict(PATCHES_PER_FRAME=96, REMOVAL_WINDOW=22, OPTIMIZATION_WINDOW=10, PATCH_LIFETIME=13,
                    KEYFRAME_THRESH=15.0, GRADIENT_BIAS=False),
    "precise": dict(PATCHES_PER_FRAME=300, REMOVAL_WINDOW=42, OPTIMIZATION_WINDOW=30, PATCH_LIFETIME=33,
                    KEYFRAME_THRESH=15.0, GRADIENT_BIAS=False),
    "fast": dict(PATCHES_PER_FRAME=48, REMOVAL_WINDOW=16, OPTIMIZATION_WINDOW=7, PATCH_LIFETIME=11,
                 KEYFRAME_THRESH=15.0, GRADIENT_BIAS=False),
}


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self.update(yaml.safe_load(f) or {})

    def merge_from_dict(self, d):
        self.update(d)

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))


def make_cfg(preset=None, **overrides):
    c = CfgNode(_DEFAULTS)
    if preset:
        c.update(PRESETS[preset])
    c.update(overrides)
    return c


cfg = make_cfg()

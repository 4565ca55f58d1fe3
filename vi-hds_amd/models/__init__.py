"""Model registry: the same string keys as the reference's models/__init__.py:19-35, so `model:` in a spec
YAML selects the same model.  Keys the HIP library does not implement raise at construction with a clear
message.  debug_constant, inducer_*, relay_* and degrader_* cannot run in the reference itself (they raise at
construction / call); they are restated from the reference's equations and labelled "parity unpinned"."""
from models import (auto_constant, debug, degrader_constant, dr_blackbox, dr_constant, inducer_constant, prpr_constant,
                    relay_constant)


class _Unsupported(object):
    def __init__(self, key, why):
        self.key, self.why = key, why

    def __call__(self, config):
        raise NotImplementedError("model '%s' is not implemented by the HIP path: %s" % (self.key, self.why))


LOOKUP = {
    "debug_constant": debug.Debug_Constant,
    "auto_constant": auto_constant.Auto_Constant,
    "auto_constant_precisions": auto_constant.Auto_Constant_Precisions,
    "degrader_constant": degrader_constant.Degrader_Constant,
    "degrader_constant_precisions": degrader_constant.Degrader_Constant_Precisions,
    "dr_constant": dr_constant.DR_Constant,
    "dr_constant_v2": dr_constant.DR_Constant_V2,
    "dr_constant_precisions": dr_constant.DR_Constant_Precisions,
    "dr_constant_precisions_v2": dr_constant.DR_Constant_Precisions_V2,
    "dr_blackbox": dr_blackbox.DR_Blackbox,
    "inducer_constant": inducer_constant.Inducer_Constant,
    "inducer_constant_precisions": inducer_constant.Inducer_Constant_Precisions,
    "prpr_constant": prpr_constant.PRPR_Constant,
    "prpr_constant_precisions": prpr_constant.PRPR_Constant_Precisions,
    "relay_constant": relay_constant.Relay_Constant,
    "relay_constant_precisions": relay_constant.Relay_Constant_Precisions,
}

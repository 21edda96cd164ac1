"""weather2alert_amd -- MI355X-native vectorised HeatAlertEnv (drop-in for weather2alert.env).

    from weather2alert_amd import HeatAlertEnv, HeatAlertVecEnv

The environment classes need a ROCm GPU and the in-tree HIP library
(`python -m weather2alert_amd.build`); importing them on a CPU-only host works, constructing
them raises. There is no CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["HeatAlertEnv", "HeatAlertVecEnv", "KernelOptions", "CompiledTables", "compile_from_files", "compile_from_synth"]


def __getattr__(name):
    if name == "HeatAlertVecEnv":
        from .env import HeatAlertVecEnv

        return HeatAlertVecEnv
    if name == "HeatAlertEnv":
        from .dropin import HeatAlertEnv

        return HeatAlertEnv
    if name == "KernelOptions":
        from .options import KernelOptions

        return KernelOptions
    if name in ("CompiledTables", "compile_from_files", "compile_from_synth", "DeviceTables"):
        from . import tables

        return getattr(tables, name)
    raise AttributeError(name)

"""yak_amd -- MI355X-native TSDF fusion engine behind Yak's API (hot path: TsdfVolume::integrate).

  csrc/       hand-written HIP kernels for gfx950 + the C ABI (include/yakhip.h) -> libyakhip.so
  capi.py     ctypes binding of that ABI (tests, bench, smoke)
  synth.py    deterministic synthetic depth scenes

The product path is the HIP library; there is no CPU fallback.
"""
__version__ = "0.1.0"

"""sfft_amd -- MI355X-native SFFT subtraction core behind the sfft operator surface.

Public names mirror the reference package for the hot path only (sfft/__init__.py:16-18):
    Customized_Packet, PureCupy_Customized_Packet, and sfft_amd.sfftcore.{SingleSFFTConfigure,
    ElementalSFFTSubtract, GeneralSFFTSubtract, GeneralSFFTSubtract_PureCupy}.
Importing the package does not need a GPU; calling any operator needs libsfft_amd.so and a gfx950 device.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: the operators import torch; keep `import sfft_amd` light for build tooling
    if name == "Customized_Packet":
        from .CustomizedPacket import Customized_Packet
        return Customized_Packet
    if name == "PureCupy_Customized_Packet":
        from .PureCupyCustomizedPacket import PureCupy_Customized_Packet
        return PureCupy_Customized_Packet
    raise AttributeError(name)

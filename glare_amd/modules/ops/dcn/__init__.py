from .deform_conv import (ModulatedDeformConv, ModulatedDeformConvFunction, ModulatedDeformConvPack, deform_conv_ext,  # noqa: F401
                          modulated_deform_conv)

__all__ = ["ModulatedDeformConv", "ModulatedDeformConvPack", "ModulatedDeformConvFunction", "modulated_deform_conv",
           "deform_conv_ext"]

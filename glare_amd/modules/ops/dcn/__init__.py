from .deform_conv import (DeformConv, DeformConvFunction, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvFunction,  # noqa: F401
                          ModulatedDeformConvPack, deform_conv, deform_conv_ext, modulated_deform_conv)

__all__ = ["DeformConv", "DeformConvPack", "DeformConvFunction", "deform_conv", "ModulatedDeformConv", "ModulatedDeformConvPack",
           "ModulatedDeformConvFunction", "modulated_deform_conv", "deform_conv_ext"]

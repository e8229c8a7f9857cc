from .deform_conv import (ModulatedDeformConv, ModulatedDeformConvPack, modulated_deform_conv)

__all__ = ['ModulatedDeformConv', 'ModulatedDeformConvPack', 'modulated_deform_conv']

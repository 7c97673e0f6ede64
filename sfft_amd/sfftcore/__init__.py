"""Mirror of sfft/sfftcore/__init__.py:7-8 (same public names)."""
from .SFFTConfigure import SingleSFFTConfigure
from .SFFTSubtract import (ElementalSFFTSubtract, ElementalSFFTSubtract_PureCupy, GeneralSFFTSubtract,
                           GeneralSFFTSubtract_PureCupy)

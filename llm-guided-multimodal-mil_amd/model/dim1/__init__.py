from .ABMIL import ABMIL      # noqa: F401
from .CLIP import CLIP        # noqa: F401

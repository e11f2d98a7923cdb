from .bitmatrix import BitMatrix
from .transpose import transpose

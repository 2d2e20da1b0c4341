"""Default data types (reference src/tike/precision.py:1-11)."""
import numpy as np

integer = np.intc
floating = np.single
cfloating = np.csingle

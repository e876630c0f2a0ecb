"""TEST INFRASTRUCTURE ONLY. Physical constants, restated from
xrt/backends/raycing/physconsts.py:5-36 (values AND the floating-point
expressions that derive CH/CHBAR, so the doubles are bit-identical)."""
PI = 3.1415926535897932384626433832795
PI2 = 6.283185307179586476925286766559
C_CM = 2.99792458e10          # physconsts.py:13
HPLANCK = 6.626069573e-27     # physconsts.py:18
EV2ERG = 1.602176565e-12      # physconsts.py:19
R0 = 2.817940285e-5           # physconsts.py:32, Angstrom
AVOGADRO = 6.02214199e23      # physconsts.py:33
CH = HPLANCK * C_CM / EV2ERG * 1e8   # physconsts.py:34-35 -> 12398.419297617678
CHBAR = CH / PI2                      # physconsts.py:36 -> 1973.2697177417986

"""Physical constants with the same floating-point derivation as
xrt/backends/raycing/physconsts.py:5-36 (CH, CHBAR must be the same doubles:
k = E/CHBAR*1e7 feeds a 4e11 rad phase)."""
PI = 3.1415926535897932384626433832795
PI2 = 6.283185307179586476925286766559
C = 2.99792458e10            # cm/s
HPLANCK = 6.626069573e-27    # erg s
EV2ERG = 1.602176565e-12
R0 = 2.817940285e-5          # A
AVOGADRO = 6.02214199e23
CHeVcm = HPLANCK * C / EV2ERG
CH = CHeVcm * 1e8            # eV A   = 12398.419297617678
CHBAR = CH / PI2             # eV A   = 1973.2697177417986
# undulator sources (physconsts.py:7-31)
SQ2 = 2**0.5
SQPI = PI**0.5
SIE0 = 1.602176565e-19
E0 = SIE0 * C / 10
M0 = 9.109383701528e-28      # g
K2B = 2 * PI * M0 * C**2 * 0.001 / E0
FINE_STR = 1 / 137.03599976
E2WC = 5067.7309392068091

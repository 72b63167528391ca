"""epik_amd: MI355X-native phylogenetic placement engine (EPIK-compatible).

Only what the `epik::placer` hot path needs lives here: the HIP kernels and the
C-ABI (`csrc/`), the ctypes binding (`capi`), the host-side mirror of the
reference's placer interface (`placer`), and the callers / data formats on either
side of the path (FASTA in, jplace out, synthetic databases for tests and bench).
"""

__version__ = "0.1.0"

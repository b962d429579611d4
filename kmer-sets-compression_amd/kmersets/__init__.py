"""Host-side Python plumbing for the MI355X k-mer-set hot path.

The product is the HIP library behind include/kmersets_hip.h (built into
kmer-sets-compression_amd/csrc/libkmersets_hip.so); this package only binds it
(ctypes, raw device pointers) and generates seeded synthetic inputs.
"""

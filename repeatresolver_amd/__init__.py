"""MI355X-native PW_ReAligner hot path (iterative sum-of-pairs MSA realignment).

Product code: csrc/ (HIP kernels + C ABI, include/pwr.h), realigner.py (ctypes mirror of the
reference's interface), datagen.py (seeded inputs), window.py / sharding.py (section split and
multi-GPU sharding).  The CPU checker lives in oracle/ and is never imported from here."""

"""The array-level model of the device SPSS encode (tests/model_encode.py) must give
the oracle's unitigs and SPSS strings, string for string and in the same order
(CPU only).  The HIP kernels implement this model; the GPU tests compare them with
the oracle directly."""
import numpy as np
import pytest

import oracle_lib as ol
from kmersets import synth
from model_encode import EncodeModel


def check(k, n, kb, kmers):
    s = ol.Set.from_kmers(k, n, kb, kmers)
    m = EncodeModel(s.kmers(), k)
    assert m.unitigs() == s.unitigs()
    assert m.spss() == s.spss()
    return m


@pytest.mark.parametrize("seq", ["AACCGTTAGCAT", "ACGTACGTACG", "AAAAAAAAA", "AACCGGTT",
                                 "ACGTTGCAACGT", "ATATATATAT", "GATTACAGATTACAGATTACA",
                                 "CCCCCCGGGGGG"])
def test_small_shapes(seq):
    check(5, 3, 1, synth.canonical_set_of_bases(synth.bases_of_string(seq), 5))


@pytest.mark.parametrize("seed", range(12))
def test_random_reads(seed):
    k = [5, 7, 9, 9, 11, 15][seed % 6]
    size = [50, 300, 2000, 4000, 3000, 3000][seed % 6]
    km = synth.random_read_kmers(k, min(size, 4 ** k // 3), seed=seed, canonical=True)
    check(k, min(10, 2 * k - 4), 4, km)


def test_loops_and_loop_cuts():
    loops = cuts = 0
    for seed in range(120):
        k = [5, 7, 9, 11][seed % 4]
        km = synth.circular_with_tails(k, 20 + (seed * 7) % 150, seed % 5, 1 + seed % 4, seed)
        m = check(k, min(10, 2 * k - 4), 4, km)
        loops += sum(1 for c in m.uclass.values() if c == 3)
        mate = m.greedy_rounds()
        cuts += m.cut_loops(mate) != list(mate)
    assert loops > 5 and cuts > 10, "the generator must exercise both serial passes of the reference"


def test_family_algebra_results():
    for k, n in ((9, 10), (15, 14), (23, 14)):
        sets = synth.phylogeny_sets(k, 3, 2500, seed=k)
        for s in sets:
            check(k, n, 4, s)
        check(k, n, 4, np.intersect1d(sets[0], sets[1]))
        check(k, n, 4, np.setdiff1d(sets[0], sets[1]))

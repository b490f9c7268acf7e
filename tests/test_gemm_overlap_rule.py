"""The rule every internal matrix product is launched under (csrc/gemm.hip: launch_gemm aborts on a result that shares an element with an operand).  Round 6: the
blocked Cholesky's panel solve wrote its result over its own operand through 32 x 32 tiles -- a race visible only on the first evaluation of a fresh process
(profiles/r06_first_evaluation_race.txt).  Host arithmetic: runs without a GPU."""
import itertools

import numpy as np

from gparml_amd import _lib


def meets(x_off, rows, cols, ld, c_off, m, n, ldc):
    return bool(_lib.load().gp_debug_operands_overlap(x_off, rows, cols, ld, c_off, m, n, ldc))


def test_the_panel_solve_in_place_is_refused_and_the_work_panel_is_not():
    NB, Mp = 128, 1024
    for j in range(7):
        rem = 7 - j
        panel = ((j + 1) * NB) * Mp + j * NB                       # L21 = A[(j+1)NB.., jNB..jNB+NB)
        assert meets(panel, rem * NB, NB, Mp, panel, rem * NB, NB, Mp)                                  # C = A: round 5
        assert not meets(panel, rem * NB, NB, Mp, 2 * Mp * Mp, rem * NB, NB, NB)                        # C = the work panel behind the matrix
        assert not meets(j * NB * Mp + j * NB, NB, NB, Mp, panel, rem * NB, NB, Mp)                     # the diagonal block of Linv's layout against the panel below it
        # trailing update: reads the panel (columns of block j), writes the blocks to the right of it in the same rows
        assert not meets(panel, rem * NB, NB, Mp, panel + NB, rem * NB, rem * NB, Mp)


def test_windows_of_one_parent_matrix_against_brute_force():
    rs = np.random.RandomState(0)
    ld, R = 40, 30
    for _ in range(4000):
        r0, c0, rows, cols = rs.randint(0, R - 8), rs.randint(0, ld - 8), rs.randint(1, 9), rs.randint(1, 9)
        r1, c1, m, n = rs.randint(0, R - 8), rs.randint(0, ld - 8), rs.randint(1, 9), rs.randint(1, 9)
        a = {(r0 + i) * ld + c0 + k for i, k in itertools.product(range(rows), range(cols))}
        b = {(r1 + i) * ld + c1 + k for i, k in itertools.product(range(m), range(n))}
        assert meets(r0 * ld + c0, rows, cols, ld, r1 * ld + c1, m, n, ld) == bool(a & b), (r0, c0, rows, cols, r1, c1, m, n)


def test_different_leading_dimensions_fall_back_to_address_ranges():
    assert not meets(0, 16, 16, 64, 16 * 64, 16, 16, 16)         # disjoint ranges
    assert meets(0, 16, 16, 64, 8, 16, 16, 16)                   # ranges intersect: refused even if no element is shared (conservative)

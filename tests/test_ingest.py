"""The library's CSV ingest (gp_csv_shape / gp_csv_read) against numpy.genfromtxt(file, delimiter=','), the call it
replaces (local_MapReduce.py:197-199, 325-327).  Host-only code: runs without a GPU."""
import time

import numpy as np
import pytest

from gparml_amd.gpu_MapReduce import _read_csv


def _gen(path):
    Y = np.genfromtxt(path, delimiter=',')
    if Y.ndim == 1:
        Y = np.atleast_2d(Y).T                      # local_MapReduce.py:198-199
    return Y


def test_matches_genfromtxt_on_a_written_shard(tmp_path):
    rs = np.random.RandomState(0)
    Y = rs.randn(5000, 7) * 10.0 ** rs.randint(-8, 8, size=(5000, 7))
    p = tmp_path / 'shard.csv'
    np.savetxt(p, Y, delimiter=',')                 # the format the reference's data sets use ('%.18e')
    got = _read_csv(str(p))
    assert got.shape == (5000, 7)
    assert np.array_equal(got, _gen(p)) and np.array_equal(got, Y)      # strtod round-trips %.18e exactly


def test_formats_blank_and_comment_lines(tmp_path):
    p = tmp_path / 'odd.csv'
    p.write_bytes(b"# header comment\n1,2.5,-3e-2\r\n\n  4 , 5e+3,.5\n   \n7,8,9")
    got = _read_csv(str(p))
    assert np.array_equal(got, np.array([[1, 2.5, -0.03], [4, 5000.0, 0.5], [7, 8, 9]]))
    assert np.array_equal(got, _gen(p))


def test_single_column_file_is_two_dimensional(tmp_path):
    p = tmp_path / 'col.csv'
    p.write_text("1.5\n2.5\n-4\n")
    got = _read_csv(str(p))
    assert got.shape == (3, 1) and np.array_equal(got, _gen(p))


def test_missing_values_are_nan(tmp_path):
    p = tmp_path / 'nan.csv'
    p.write_text("1,,3\n4,abc,6\nnan,inf,-inf\n")
    got = _read_csv(str(p))
    ref = _gen(p)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])


def test_errors_map_to_assertion_error(tmp_path):
    p = tmp_path / 'ragged.csv'
    p.write_text("1,2,3\n4,5\n")
    with pytest.raises(AssertionError):
        _read_csv(str(p))
    with pytest.raises(AssertionError):
        _read_csv(str(tmp_path / 'does_not_exist.csv'))


def test_parallel_parse_of_a_larger_file(tmp_path):
    rs = np.random.RandomState(1)
    Y = rs.randn(200000, 20)
    p = tmp_path / 'big.csv'
    np.savetxt(p, Y, delimiter=',', fmt='%.17g')
    t = time.time()
    got = _read_csv(str(p))
    dt = time.time() - t
    assert np.array_equal(got, Y)
    assert dt < 30.0                                # numpy.genfromtxt needs ~12 s for these 4e6 numbers on one core

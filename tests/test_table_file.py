"""Reference table-file interchange: the C-ABI reader/writer against the oracle's byte-level
restatement of the cereal layout.  Host only -- runs without a GPU."""
import numpy as np
import pytest

import gvi_oracle as o
from gaussianvi_amd import api

KEYS = [(1, 10), (4, 3), (2, 5), (5, 2), (1, 1)]


def test_writer_matches_byte_image_of_oracle_tables(tmp_path):
    path = str(tmp_path / "SparseGHQuadratureWeights_cereal.bin")
    api.table_file_write(path, KEYS)
    want = o.cereal_table_bytes([(d, p) + tuple(o.nwspgr(d, p)) for d, p in KEYS])
    got = open(path, "rb").read()
    assert len(got) == len(want)
    tabs = o.cereal_table_parse(got)
    for d, p in KEYS:
        Z, w = o.nwspgr(d, p)
        gz, gw = tabs[(float(d), float(p))]
        assert np.array_equal(gz, Z)                       # nodes are table constants: bit-exact
        assert np.abs(gw - w).max() < 1e-13
    assert [(a, b) for a, b, _ in api.table_file_list(path)] == [(float(d), float(p)) for d, p in KEYS]


def test_reader_finds_keys_in_any_order_and_reports_missing(tmp_path):
    """unordered_map iteration order is unspecified: shuffle the entries, mix in an off-grid key."""
    rng = np.random.default_rng(3)
    entries = [(d, p) + tuple(o.nwspgr(d, p)) for d, p in KEYS]
    entries.append((3.0, 2.5, rng.normal(size=(7, 3)), rng.normal(size=7)))      # key (3, 2.5): not an int degree
    order = rng.permutation(len(entries))
    path = str(tmp_path / "t.bin")
    open(path, "wb").write(o.cereal_table_bytes([entries[i] for i in order]))
    for d, p in KEYS:
        Z, w = api.table_file_read(path, d, p)
        Zo, wo = o.nwspgr(d, p)
        assert np.array_equal(Z, Zo) and np.array_equal(w, wo)
    lst = api.table_file_list(path)
    assert len(lst) == len(entries) and (3.0, 2.5, 7) in lst
    with pytest.raises(api.GviError) as e:
        api.table_file_read(path, 12, 5)
    assert e.value.status == 4                              # GVI_ERR_NOTABLE
    with pytest.raises(api.GviError):
        api.table_file_read(str(tmp_path / "absent.bin"), 1, 10)
    open(path, "ab").close()
    trunc = str(tmp_path / "trunc.bin")
    open(trunc, "wb").write(open(path, "rb").read()[:-9])
    with pytest.raises(api.GviError):
        api.table_file_list(trunc)


def test_write_rejects_untabulated_key(tmp_path):
    with pytest.raises(api.GviError) as e:
        api.table_file_write(str(tmp_path / "x.bin"), [(2, 26)])
    assert e.value.status == 4

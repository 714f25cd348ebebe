"""DESIGN.md section 4's kernel table must come out of profiles/ MECHANICALLY (round 5's table could not be reproduced from the
JSON it cited: two entries of r05_pmc_all.json were byte-identical copies of entries measured on a kernel that no longer
shipped).  No GPU needed:
  * no two differently named entries of the counter record DESIGN cites are equal (a stale copy cannot hide),
  * profiles/r06_kernel_table.md IS the output of tools/pmc_table.py on profiles/r06_pmc_all.json,
  * DESIGN.md includes that file verbatim between its markers,
  * every number of the table follows from the entry's own fields (counters, clock probe) -- recomputed here to 2 %."""
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = os.path.join(ROOT, "profiles", "r06_pmc_all.json")
TABLE = os.path.join(ROOT, "profiles", "r06_kernel_table.md")
BEGIN, END = "<!-- kernel-table: profiles/r06_kernel_table.md, verbatim -->", "<!-- /kernel-table -->"


def _pmc_table():
    spec = importlib.util.spec_from_file_location("pmc_table", os.path.join(ROOT, "tools", "pmc_table.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def record():
    if not os.path.exists(REC):
        pytest.skip("profiles/r06_pmc_all.json not committed yet")
    return json.load(open(REC))


def test_no_two_entries_of_the_counter_record_are_copies(record):
    names = sorted(record)
    assert len(names) >= 14, names
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            ca, cb = record[a]["counters_per_dispatch"], record[b]["counters_per_dispatch"]
            assert ca != cb, "entries %s and %s hold the same counters: one of them is a stale copy" % (a, b)
            assert record[a].get("clock_probe") != record[b].get("clock_probe") or record[a].get("clock_probe") is None, (a, b)
    # every entry was measured in this round's run (the record carries no entry from an older file)
    for k, v in record.items():
        assert v.get("round") == "r06", (k, v.get("round"))


def test_the_table_file_is_the_scripts_output_and_design_includes_it(record):
    want = "\n".join(_pmc_table().table(record, True)) + "\n"
    assert open(TABLE).read() == want, "profiles/r06_kernel_table.md is not `python tools/pmc_table.py profiles/r06_pmc_all.json`"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert BEGIN in design and END in design
    block = design[design.index(BEGIN) + len(BEGIN):design.index(END)].strip("\n") + "\n"
    assert block == want, "DESIGN.md's kernel table differs from profiles/r06_kernel_table.md"


def test_every_cell_follows_from_the_entrys_own_fields(record):
    rows = [l for l in open(TABLE).read().splitlines()[2:] if l.startswith("|")]
    m = _pmc_table()
    by_name = {m.NAMES[k]: k for k in m.ORDER if k in record}
    assert len(rows) == len(by_name) >= 14
    for line in rows:
        cells = [c.strip().strip("*") for c in line.strip("|").split("|")]
        v = record[by_name[cells[0]]]
        c, cp = v["counters_per_dispatch"], v["clock_probe"]
        units_c = v["units_per_dispatch"]
        valu_pu, lds_pu = c["SQ_INSTS_VALU"] / units_c, c["SQ_INSTS_LDS"] / units_c
        a, b = [float(x) for x in cells[2].split("/")]
        assert abs(a - valu_pu) <= 0.02 * valu_pu + 0.006 and abs(b - lds_pu) <= 0.02 * lds_pu + 0.006, (cells[0], cells[2])
        traffic = (2.0 * c["FETCH_SIZE"] * 32.0 + c["WRITE_SIZE"] * 32.0) / units_c if False else None   # (units differ by counter: use derived)
        assert abs(float(cells[3]) - v["derived"]["hbm_over_algorithmic"]) <= 1e-4
        ghz = cp["shader_clock_MHz_median"] / 1e3
        assert abs(float(cells[4].split()[0]) - ghz) <= 0.02 * ghz
        rate = cp["units"] / (cp["ms_last_launch_events"] * 1e-3)
        assert abs(float(cells[5]) - rate) <= 0.02 * rate
        valu = valu_pu * cp["units"] * 4.0 / (4.0 * 256 * ghz * 1e9 * cp["ms_last_launch_events"] * 1e-3)
        hbm = rate * v["alg_bytes_per_unit"] / 8e12
        assert abs(float(cells[6]) - valu) <= 0.02 * valu + 0.006 and abs(float(cells[7]) - hbm) <= 0.02 * hbm + 0.006, cells

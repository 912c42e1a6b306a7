"""cpp/NAM/multi_device.h on the CPU: the device-list parser and the dealing of files to devices, through
`render --plan-only` (no GPU is touched; --device-count stands in for the visible devices)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, model_path

EXE = os.path.join(ROOT, "cpp", "tools", "render")


@pytest.fixture(scope="module")
def render_exe(nam_lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cpp")], stdout=subprocess.DEVNULL)
    return EXE


def _wav(path, n):
    data = np.zeros(n, dtype="<f4").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 48000, 48000 * 4, 4, 32)
                + b"data" + struct.pack("<I", len(data)) + data)


def _plan(render_exe, tmp_path, devices, count, lengths):
    files = []
    for i, n in enumerate(lengths):
        p = str(tmp_path / f"f{i}.wav")
        _wav(p, n)
        files.append(p)
    r = subprocess.run([render_exe, "--devices", devices, "--device-count", str(count), "--plan-only", model_path("wavenet"), "--batch",
                        str(tmp_path / "out")] + files, capture_output=True, text=True, timeout=60)
    plan = {}
    for line in r.stdout.splitlines():
        head, rest = line.split(": ", 1)
        dev = int(head.split()[1])
        batch = int(head.split("batch ")[1].rstrip(")"))
        plan.setdefault((batch, dev), []).append(int(rest.rsplit("(", 1)[1].split()[0]))
    return r, plan


def test_files_are_dealt_longest_first_in_snake_order(render_exe, tmp_path):
    lengths = [100, 900, 300, 800, 200, 700, 400, 600, 500]
    r, plan = _plan(render_exe, tmp_path, "0-2", 8, lengths)
    assert r.returncode == 0, r.stderr
    # sorted 900 800 700 | 600 500 400 (reversed) | 300 200 100
    in_file_order = lambda vals: sorted(vals, key=lengths.index)  # a device lists its files in command-line order
    assert plan == {(0, 0): in_file_order([900, 400, 300]), (1, 1): in_file_order([800, 500, 200]),
                    (2, 2): in_file_order([700, 600, 100])}
    counts = [len(v) for v in plan.values()]
    assert max(counts) - min(counts) <= 1 and sum(counts) == len(lengths)


@pytest.mark.parametrize("spec,count,want", [("all", 4, [0, 1, 2, 3]), ("0-7", 8, list(range(8))), ("0,2,5", 8, [0, 2, 5]),
                                              ("1-2,6", 8, [1, 2, 6]), ("0,0", 1, [0, 0]), ("3", 4, [3])])
def test_device_list_forms(render_exe, tmp_path, spec, count, want):
    r, plan = _plan(render_exe, tmp_path, spec, count, [10 * (i + 1) for i in range(2 * len(want))])
    assert r.returncode == 0, r.stderr
    assert [dev for (batch, dev) in sorted(plan)] == want


@pytest.mark.parametrize("spec,count", [("0-8", 8), ("4", 4), ("2-1", 8), ("a", 8), ("0,,1", 8), ("all", 0)])
def test_bad_device_lists_are_errors(render_exe, tmp_path, spec, count):
    r, _ = _plan(render_exe, tmp_path, spec, count, [10, 20])
    assert r.returncode == 1 and "device list" in r.stderr

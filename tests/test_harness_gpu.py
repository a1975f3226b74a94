"""GPU: the two CLIs of the reference harness (run_latency_attention.py / run_latency_kernel.py at the repo root) run end to
end with the reference's flags and print the reference's result line (run_latency_attention.py:106) -- Palu leg, dense
baseline leg (build_attention, :29-38), packed cache, and the head-group-parallel leg under torch.distributed.run."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
LINE = re.compile(r"Finished, prompt_len: (\d+), latency: ([0-9.]+) milliseconds")


def _run(args, timeout=600):
    r = subprocess.run(args, cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _json_line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_attention_harness_palu_and_dense_legs():
    base = [sys.executable, "run_latency_attention.py", "--prompt_len", "4096", "--repeats", "10", "--json"]
    palu = _run(base + ["--palu", "--fast_init", "--rank_k", "1024", "--rank_v", "3072", "--group_size", "4", "--cache_graph"])
    dense = _run(base)
    dense_graph = _run(base + ["--cache_graph"])        # the dense leg must survive a graph capture too
    assert LINE.search(dense_graph)
    for out in (palu, dense):
        m = LINE.search(out)
        assert m and int(m.group(1)) == 4096 and float(m.group(2)) > 0
    jp, jd = _json_line(palu), _json_line(dense)
    assert jd["attention"] == "dense" and jp["rank_k"] == 1024
    assert jp["latency_us"] < jd["latency_us"]            # the point of the comparison the reference CLI makes


def test_attention_harness_packed_cache():
    out = _run([sys.executable, "run_latency_attention.py", "--palu", "--fast_init", "--rank_k", "1024", "--rank_v", "3072",
                "--prompt_len", "4096", "--repeats", "5", "--bits", "3", "--hadamard", "--json"])
    assert LINE.search(out) and _json_line(out)["bits"] == 3


def test_attention_harness_head_parallel_leg_under_torchrun():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29533", "run_latency_attention.py", "--palu", "--gpus", "1", "--rank_k", "1024", "--rank_v",
                "3072", "--prompt_len", "4096", "--repeats", "10", "--cache_graph", "--json"])
    assert LINE.search(out) and "gpus=1" in out
    j = _json_line(out)
    assert j["gpus"] == 1 and j["cache_graph"] is True and j["latency_us"] > 0


def test_kernel_harness(tmp_path):
    """the three providers of kernel/abx_rope.py:180-221 (WX, Torch, Ours) and the CSV of :227-228"""
    import csv
    import json
    out = _run([sys.executable, "run_latency_kernel.py", "--total_rank", "1024", "--group_size", "4", "--target_seq_lens", "4096",
                "--json", "--save_path", str(tmp_path)])
    assert "4096" in out
    rows = json.loads([l for l in out.splitlines() if l.startswith("[")][-1])
    assert rows[0]["torch_us"] > rows[0]["ours_us"] > 0 and rows[0]["WX_us"] > 0
    with open(tmp_path / "low-rank-rank-1024-group-8.csv") as f:
        got = list(csv.reader(f))
    assert got[0][:4] == ["seq_len", "WX", "Torch", "Ours"] and got[1][0] == "4096"

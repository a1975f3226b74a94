"""The sharded decode step (HeadParallelDecoder.step, what `bench.py --gpus N` runs) with world_size 2 on ONE GPU:
both ranks on cuda:0, the all-gather carried by gloo through the host (RCCL refuses two ranks on one device).
Sharded output == unsharded output."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_unsharded():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hp_two_ranks_one_gpu.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max|diff|" in r.stdout


@pytest.mark.parametrize("oproj", ["sharded", "replicated"])
def test_captured_two_rank_step_equals_eager_and_unsharded(oproj):
    """the N > 1 step as ONE hipGraph.  RCCL refuses two ranks on one device and gloo cannot be
    captured, so both ranks' decoders live on cuda:0 and exchange through `LocalExchange` -- plain device ops behind the
    same two-method interface as the RCCL collective (head_parallel.DistExchange), hence part of the captured graph.
    Replay == eager == the unsharded single-GPU step; a replay with a new token (static input buffer) follows it."""
    import torch
    from palu_amd.kernel import head_parallel as hp
    H, G, D, HID, Rk, Rv, LP = 32, 8, 128, 1024, 128, 384, 3000
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    full = {"wq": (torch.randn(H * D, HID, device=dev) / 32).half(), "vt_k": (torch.randn(G * Rk, HID, device=dev) / 32).half(),
            "vt_v": (torch.randn(G * Rv, HID, device=dev) / 32).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": (torch.randn(HID, H * Rv, device=dev) * 0.02).half()}
    cap = LP + 64
    k_all = torch.randn(G, cap, Rk, device=dev, dtype=torch.float16)
    v_all = torch.randn(G, cap, Rv, device=dev, dtype=torch.float16)
    tok = torch.randn(HID, device=dev, dtype=torch.float16)

    def decoders(world, exchange):
        out = []
        for r in range(world):
            plan = hp.make_plan(world, r, H, G, D, Rk, Rv)
            w = {k: (v.contiguous() if k != "wo" else v) for k, v in hp.shard_weights(plan, full, oproj=oproj).items()}
            kc, vc = hp.shard_cache(plan, k_all.clone(), v_all.clone())
            out.append(hp.HeadParallelDecoder(plan, w, kc.contiguous(), vc.contiguous(), HID, exchange=exchange))
        return out
    ref = decoders(1, None)[0].step(tok, LP, LP).clone()
    ex = hp.LocalExchange(2)
    ranks = decoders(2, ex)
    for r, d in enumerate(ranks):
        ex.register(r, partial=d.partial, ctx=d.ctx)
        assert d.oproj_sharded == (oproj == "sharded")

    def both():
        for d in ranks:
            d.step_local(tok, LP, LP)
        return [d.step_finish() for d in ranks]
    eager = [o.clone() for o in both()]
    torch.cuda.synchronize()
    for o in eager:
        torch.testing.assert_close(o.float(), ref.float(), rtol=1e-3, atol=1e-3)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        both()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        outs = both()
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    for o, e in zip(outs, eager):
        assert torch.equal(o, e)                       # the captured step is the eager step, exchange included
    # new token through the static input buffer: the replay follows it
    tok2 = torch.randn(HID, device=dev, dtype=torch.float16)
    ref2 = decoders(1, None)[0].step(tok2, LP, LP).clone()
    tok.copy_(tok2)
    g.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(outs[0].float(), ref2.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_p2p_exchange_between_processes(world):
    """The one-shot peer-to-peer exchange (csrc/exchange.hip, IpcExchange): N processes on this GPU map each other's buffers
    through hipIpc handles (uncached device memory); 200 checked all-gathers (3 KiB) and fp32 all-reduces (16 KiB), graph replay,
    no timed-out wait; world 8 = the rank count of BASELINE config 5."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_two_procs.py"), "--world", str(world)], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EXCHANGE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_single_rank_capture_method():
    """HeadParallelDecoder.capture(): warm-up + capture + replay of the whole step (world 1: no collective)."""
    import torch
    from palu_amd.kernel import head_parallel as hp
    H, G, D, HID, Rk, Rv, LP = 32, 8, 128, 512, 128, 384, 700
    dev = torch.device("cuda", 0)
    torch.manual_seed(9)
    full = {"wq": (torch.randn(H * D, HID, device=dev) / 16).half(), "vt_k": (torch.randn(G * Rk, HID, device=dev) / 16).half(),
            "vt_v": (torch.randn(G * Rv, HID, device=dev) / 16).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": (torch.randn(HID, H * Rv, device=dev) * 0.02).half()}
    plan = hp.make_plan(1, 0, H, G, D, Rk, Rv)
    kc = torch.randn(G, LP + 64, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(G, LP + 64, Rv, device=dev, dtype=torch.float16)
    tok = torch.randn(HID, device=dev, dtype=torch.float16)
    dec = hp.HeadParallelDecoder(plan, full, kc, vc, HID)
    eager = dec.step(tok, LP, LP).clone()
    replay = dec.capture(tok, LP, LP)
    dec.out.zero_()
    replay()
    torch.cuda.synchronize()
    assert torch.equal(dec.out, eager)

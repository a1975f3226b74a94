"""CPU oracle for the Palu low-rank-KV decode path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``palu_amd/`` may import this package.
The only permitted users are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the timed
CPU baseline, never as a compute fallback for the HIP path.

Parity status: PINNED.  Every function in ``palu_oracle`` restates a reference
function (file:line cited in its docstring) and is checked in
``tests/test_oracle_golden.py`` against golden vectors produced by importing the
reference itself (``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` is mounted).  Exception: the external CUDA op
``fast_hadamard_transform.hadamard_transform`` is an un-vendored submodule with no
recoverable pin ("parity unpinned" at that one boundary); the oracle follows the
in-tree pure-torch ``matmul_hadU`` instead, which the reference states is equivalent.
"""
from .palu_oracle import *  # noqa: F401,F403

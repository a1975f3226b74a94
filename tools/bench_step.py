#!/usr/bin/env python3
"""Quick per-kernel timing of the decode step (no CPU baseline)."""
import subprocess, sys
sys.exit(subprocess.call([sys.executable, "bench.py", "--no_cpu_baseline", "--steps", "100"] + sys.argv[1:]))

"""BASELINE configs[4] alone: 10k queries x 100k entities, top-100 (bench.query_bench)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sert_amd import _capi  # noqa: E402

if __name__ == '__main__':
    print(bench.query_bench(_capi, cpu=False, reps=5))

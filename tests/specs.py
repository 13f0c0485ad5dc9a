"""Benchmark / fixture network specs shared by the tests, bench.py and the golden generator (kraken_amd/specs.py)."""
from kraken_amd.specs import BENCH_A, BENCH_A_RGB, BENCH_B, bench_codec  # noqa: F401

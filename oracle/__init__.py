"""ORACLE -- test infrastructure only.

CPU restatements of the reference algorithms on the IndexTTS inference hot path, each function citing the reference
file:line it follows.  Pinned against fixtures produced by running the reference itself (tests/golden/).  Nothing in
the product package (index-tts-lora_amd/) imports from here; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker / the timed CPU baseline.
"""

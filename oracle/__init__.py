"""CPU oracle for the SA-M4C hot path — TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline.  The product path
(`sam-textvqa_amd/`) never imports it and never falls back to it.
"""

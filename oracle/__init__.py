"""CPU oracle for the BIGSI query hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (bigsi_amd/) never does.  Parity pin: tests/test_oracle_golden.py checks every
function here against vectors produced by running the unmodified reference
(tests/golden/make_golden.py).
"""

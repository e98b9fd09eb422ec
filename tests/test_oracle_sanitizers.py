"""CPU tier: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference
itself builds with -Wall -Wextra only).  oracle/oracle_kat.cpp drives every oracle entry point -- the five
reference parameter sets and three more on six ragged sizes, threads, diagnostics, the analysis rows on a field
with NaN / Inf / 3e38, the display path -- and prints one hash per result.  The sanitized build must run clean and
print the hashes of the normal -O3 build: no result rests on undefined behaviour or on the optimisation level.
Host code only (the GPU sanitizers are not available on this pool)."""
import os
import subprocess

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_runs_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", ORACLE, "oracle_kat", "oracle_kat_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    san = subprocess.run([os.path.join(ORACLE, "oracle_kat_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert san.returncode == 0, san.stdout[-2000:] + san.stderr[-4000:]
    assert "runtime error" not in san.stderr and "AddressSanitizer" not in san.stderr, san.stderr[-4000:]
    ref = subprocess.run([os.path.join(ORACLE, "oracle_kat")], capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0
    assert san.stdout == ref.stdout and san.stdout.strip().endswith("oracle_kat: ok") and san.stdout.count("\n") > 60

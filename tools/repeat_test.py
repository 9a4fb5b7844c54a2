"""Run a pytest selection over and over in ONE process (state accumulates as in the whole suite) and keep the failures' messages.
GPU box: python tools/repeat_test.py N out.txt <pytest args...>"""
import io, sys, contextlib, re
import pytest
n, out = int(sys.argv[1]), sys.argv[2]
bad = 0
with open(out, "w") as f:
    for i in range(n):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = pytest.main(["-q", "-p", "no:cacheprovider", "-x"] + sys.argv[3:])
        if rc != 0:
            bad += 1
            msg = [l[:3000] for l in buf.getvalue().splitlines() if re.search(r"AssertionError|^FAILED|side differs", l)]
            f.write(f"== iteration {i}: rc {rc}\n" + "\n".join(msg) + "\n"); f.flush()
        if i % 10 == 9:
            print(f"... {i + 1} iterations, {bad} failed", flush=True)
    f.write(f"{bad} of {n} iterations failed\n")
print(f"{bad} of {n} iterations failed")

"""Build the oracle's C restatement (oracle/ssq_ref.c) into oracle/_build/libssq_ref.so with gcc.
Test infrastructure only (see the header of ssq_ref.c)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ssq_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libssq_ref.so")


def build_ref(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        cmd = ["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed: {' '.join(cmd)}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_ref(force=True))

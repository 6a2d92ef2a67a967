"""Are the kernels of the working tree instruction-for-instruction the kernels of an earlier commit?

    python profiles/isa_identity_check.py <git-rev> [file.hip ...]

Compiles every csrc/*.hip (or the files named) of <git-rev> and of the working tree for gfx950 (device code only, -S), strips labels and
comments, and prints per kernel SAME / DIFF / NEW / GONE.  Used to show that code written without a device at hand (new template
instantiations, new entry points, bigger argument structs) left the measured kernels untouched.  Needs only hipcc - no GPU."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lightning-pose_amd", "csrc")
HIPCC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")


def kernels(asm: str) -> dict[str, list[str]]:
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN2lp\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur and line.startswith("\t") and not line.strip().startswith((".", ";")):
            ins = re.sub(r"\.LBB\d+_", ".LBB_", line.split(";")[0].strip())
            out[cur].append(ins)
            if ins == "s_endpgm":
                cur = None
    return out


def compile_to_asm(src_text: str, common_text: str, workdir: str, tag: str) -> str:
    with open(os.path.join(workdir, "lp_common.h"), "w") as fh:
        fh.write(common_text.replace('"../../include/lp_hip.h"', f'"{os.path.join(ROOT, "include", "lp_hip.h")}"'))
    src = os.path.join(workdir, f"{tag}.hip")
    with open(src, "w") as fh:
        fh.write(src_text)
    out = os.path.join(workdir, f"{tag}.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out], check=True,
                   capture_output=True)
    return open(out).read()


def git_show(rev: str, path: str) -> str | None:
    res = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{path}"], capture_output=True, text=True)
    return res.stdout if res.returncode == 0 else None


def main() -> None:
    rev = sys.argv[1]
    files = sys.argv[2:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    changed = 0
    for f in files:
        rel = f"lightning-pose_amd/csrc/{f}"
        old_src = git_show(rev, rel)
        new_src = open(os.path.join(CSRC, f)).read()
        if old_src is None:
            print(f"{f}: not in {rev} (new file)")
            continue
        with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
            old = kernels(compile_to_asm(old_src, git_show(rev, "lightning-pose_amd/csrc/lp_common.h"), a, "k"))
            new = kernels(compile_to_asm(new_src, open(os.path.join(CSRC, "lp_common.h")).read(), b, "k"))
        renamed = {k: k2 for k in old if k not in new for k2 in new if k2 not in old and new[k2] == old[k]}  # e.g. a function made a template
        for k in sorted(set(old) | set(new)):
            if k in renamed:
                print(f"SAME {len(old[k]):6d}  {f}: {k[:70]}  (now {renamed[k][:60]})")
                continue
            if k in renamed.values():
                continue
            state = "NEW " if k not in old else "GONE" if k not in new else "SAME" if old[k] == new[k] else "DIFF"
            changed += state == "DIFF"
            print(f"{state} {len(new.get(k, old.get(k, []))):6d}  {f}: {k[:100]}")
    print(f"{changed} kernel(s) differ from {rev}")


if __name__ == "__main__":
    main()

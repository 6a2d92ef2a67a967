"""Per-layer table of the convolution launches of one step: forward vs data gradient vs weight gradient, from the per-launch dump
that bench.py writes when LP_DUMP_LAUNCHES=<file> is set (HIP events around every convolution C-ABI call of the last sampled step).

    LP_DUMP_LAUNCHES=gpurun_out/launches.json python bench.py --steps 5 --no-cpu-baseline     # on the GPU box
    python profiles/layer_table.py gpurun_out/launches.json                                    # anywhere

Round 6: every launch record carries the NAME of the layer it belongs to (engine._timed(layer=...): the parameter name, e.g.
backbone.5.0.conv1), and the table groups by it.  Until round 5 the table matched launches to layers by their ORDER in the step, and the
round-5 reorder of a layer's first block (the projection shortcut's data gradient before conv1's) silently swapped four rows' labels
(VERDICT r5 weak #9).  A layer with several launches of one kind (the four parity classes of a strided data gradient, the two passes of a
step that could not be joined) shows their sum; the launch count is printed when it is not 1."""
import json
import sys

BLOCKS = (3, 4, 6, 3)


def short(name: str) -> str:
    """backbone.5.0.conv1 -> l2.0.c1, backbone.5.0.downsample.0 -> l2.0.down, backbone.0 -> stem"""
    p = name.split(".")
    if p[0] != "backbone" or len(p) < 4:
        return "stem" if name == "backbone.0" else name
    kind = {"conv1": "c1", "conv2": "c2", "conv3": "c3", "downsample": "down"}[p[3]]
    return f"l{int(p[1]) - 3}.{p[2]}.{kind}"


def order():
    out = []
    for li, nb in enumerate(BLOCKS):
        for b in range(nb):
            out += [f"l{li + 1}.{b}.c1", f"l{li + 1}.{b}.c2"] + ([f"l{li + 1}.{b}.down"] if b == 0 else []) + [f"l{li + 1}.{b}.c3"]
    return out


def main():
    launches = json.load(open(sys.argv[1]))
    if not launches or len(launches[0]) < 5 or not any(x[4] for x in launches):
        sys.exit("this dump has no layer names (written before round 6): regenerate it with LP_DUMP_LAUNCHES")
    table: dict[str, dict[str, list[float]]] = {}
    for tag, gflop, us, _mb, layer in launches:
        if not layer.startswith("backbone."):
            continue
        kind = "wgrad" if "wgrad" in tag else "dgrad" if "dgrad" in tag else "fwd"
        rec = table.setdefault(short(layer), {}).setdefault(kind, [0.0, 0.0, 0])
        rec[0] += gflop
        rec[1] += us
        rec[2] += 1
    print(f"{'layer':11s} {'GFLOP':>7s} | {'fwd us':>7s} {'TF/s':>5s} | {'dgrad us':>8s} {'TF/s':>5s} {'x fwd':>5s} | {'wgrad us':>8s} {'TF/s':>5s}")
    tot = [0.0, 0.0, 0.0]
    for n in order():
        r = table.get(n, {})
        f, d, w = r.get("fwd", [0, 0, 0]), r.get("dgrad", [0, 0, 0]), r.get("wgrad", [0, 0, 0])
        gf = f[0] or w[0] or d[0]
        tot = [tot[0] + f[1], tot[1] + d[1], tot[2] + w[1]]
        cell = lambda rec: (f"{rec[1]:8.1f} {gf / rec[1] * 1e3:5.0f}" if rec[1] else f"{'-':>8s} {'-':>5s}") + (f"({rec[2]})" if rec[2] > 1 else "")  # noqa: E731
        ratio = f"{d[1] / f[1]:5.2f}" if f[1] and d[1] else f"{'-':>5s}"
        print(f"{n:11s} {gf:7.1f} | {cell(f)[1:]} | {cell(d)} {ratio} | {cell(w)}")
    stem = table.get("stem", {})
    if stem:
        print("stem        " + "  ".join(f"{k} {v[1]:.1f} us" for k, v in stem.items()))
    print(f"totals (ms): forward {tot[0] / 1e3:.2f}  data gradient {tot[1] / 1e3:.2f}  weight gradient {tot[2] / 1e3:.2f}")


if __name__ == "__main__":
    main()

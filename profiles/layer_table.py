"""Per-layer table of the convolution launches of one step: forward vs data gradient vs weight gradient, from the per-launch dump
that bench.py writes when LP_DUMP_LAUNCHES=<file> is set (HIP events around every convolution C-ABI call of the last sampled step).

    LP_DUMP_LAUNCHES=gpurun_out/launches.json python bench.py --steps 5 --no-cpu-baseline     # on the GPU box
    python profiles/layer_table.py gpurun_out/launches.json [pass]                             # anywhere

A step holds two forward/backward passes (64 labeled frames, then 128 unlabeled); `pass` = 0 / 1 picks one (default: the larger).
The launch order is the engine's: forward = stem, then per block conv1, conv2, [downsample], conv3; backward = blocks in reverse,
per block conv3, conv2, conv1, [downsample]; each data gradient is followed by its layer's weight gradient."""
import json
import sys

BLOCKS = (3, 4, 6, 3)


def names_forward():
    out = ["stem"]
    for li, nb in enumerate(BLOCKS):
        for b in range(nb):
            out += [f"l{li + 1}.{b}.c1", f"l{li + 1}.{b}.c2"] + ([f"l{li + 1}.{b}.down"] if b == 0 else []) + [f"l{li + 1}.{b}.c3"]
    return out


def names_backward():
    out = []
    for li in range(len(BLOCKS) - 1, -1, -1):
        for b in range(BLOCKS[li] - 1, -1, -1):
            out += [f"l{li + 1}.{b}.c3", f"l{li + 1}.{b}.c2", f"l{li + 1}.{b}.c1"] + ([f"l{li + 1}.{b}.down"] if b == 0 else [])
    return out


def main():
    launches = json.load(open(sys.argv[1]))
    fw = [x for x in launches if ("fwd" in x[0] or "stem>" in x[0]) and "wgrad" not in x[0]]
    dg = [x for x in launches if "dgrad" in x[0]]
    wg = [x for x in launches if "wgrad" in x[0] and "stem" not in x[0]]
    nf, nb = len(names_forward()), len(names_backward())
    if len(fw) == nf and len(dg) == nb:              # joint labeled + unlabeled pass (round 2 default): ONE forward / backward per step
        fw, dg, wg = fw * 2, dg * 2, wg * 2
    assert len(fw) == 2 * nf and len(dg) == 2 * nb, (len(fw), len(dg), "unexpected launch list: not a ResNet-50 step?")
    passes_f = [fw[:nf], fw[nf:]]
    passes_d = [dg[:nb], dg[nb:]]
    head_w = len(wg) // 2 - nb                      # the head's ConvTranspose layers come first in each backward pass
    passes_w = [wg[head_w:len(wg) // 2], wg[len(wg) // 2 + head_w:]]
    # the two passes run forward, forward, then backward in reverse order of the losses: match passes by their FLOPs
    which = int(sys.argv[2]) if len(sys.argv) > 2 else max((0, 1), key=lambda i: passes_f[i][1][1])
    f = dict(zip(names_forward(), passes_f[which]))
    big_b = max((0, 1), key=lambda i: passes_d[i][0][1]) if which == max((0, 1), key=lambda i: passes_f[i][1][1]) else \
        min((0, 1), key=lambda i: passes_d[i][0][1])
    d = dict(zip(names_backward(), passes_d[big_b]))
    w = dict(zip(names_backward(), passes_w[big_b]))
    print(f"{'layer':11s} {'GFLOP':>7s} | {'fwd us':>7s} {'TF/s':>5s} | {'dgrad us':>8s} {'TF/s':>5s} {'x fwd':>5s} | {'wgrad us':>8s} {'TF/s':>5s}")
    tot = [0.0, 0.0, 0.0]
    for n in names_forward()[1:]:
        gf, fu, du, wu = f[n][1], f[n][2], d[n][2], w[n][2]
        tot = [tot[0] + fu, tot[1] + du, tot[2] + wu]
        print(f"{n:11s} {gf:7.1f} | {fu:7.1f} {gf / fu * 1e3:5.0f} | {du:8.1f} {gf / du * 1e3:5.0f} {du / fu:5.2f} | {wu:8.1f} {gf / wu * 1e3:5.0f}")
    print(f"totals (ms): forward {tot[0] / 1e3:.2f}  data gradient {tot[1] / 1e3:.2f}  weight gradient {tot[2] / 1e3:.2f}")


if __name__ == "__main__":
    main()

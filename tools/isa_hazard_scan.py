"""Static scan of a gfx950 assembly listing (hipcc -S --cuda-device-only) for MFMA-result -> VALU-read distances.

    python tools/isa_hazard_scan.py file.s [kernel-name-substring]

For every kernel: every VALU instruction that reads a VGPR last written by a v_mfma, with the number of wait states
(instructions issued by the wave in between; s_nop N counts N + 1) since that MFMA.  LLVM's hazard recogniser
(GCNHazardRecognizer::checkMAIVALUHazards) requires passes + 2 (+ 1 on gfx950) wait states between an XDL op's VGPR
write and a VALU read of it: 11 for the 8-pass v_mfma_f32_16x16x32_f16.  The scan is linear (branches are not followed:
a distance measured across a label is a lower bound on nothing - such pairs are flagged `xlabel`)."""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(lines):
    issue = 0
    last_mfma = {}          # vgpr -> (issue index, label epoch)
    epoch = 0
    rows = []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.endswith(":") or re.match(r"^[.\w$]+:", t):
            epoch += 1
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        args = t[len(op):].split(",")
        if op == "s_nop":
            issue += int(args[0]) + 1
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            for r in regs(args[0]):
                last_mfma[r] = (issue, epoch)
            issue += 1
            continue
        if op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
            is_store = ("store" in op) or op.startswith("ds_write")
            srcs = args if is_store else args[1:]
            dst = [] if is_store else regs(args[0])
            for a in srcs:
                for r in regs(a):
                    if r in last_mfma:
                        d = issue - last_mfma[r][0] - 1
                        rows.append((op, d, last_mfma[r][1] != epoch, t))
                        break
                else:
                    continue
                break
            for r in dst:
                last_mfma.pop(r, None)
            for a in srcs:
                pass
        issue += 1
    return rows


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    for si, s0 in enumerate(starts):
        name = text[s0].split(":")[0]
        if want not in name:
            continue
        s1 = starts[si + 1] if si + 1 < len(starts) else len(text)
        body = text[s0 + 1:s1]
        end = next((i for i, l in enumerate(body) if l.startswith(".Lfunc_end")), len(body))
        rows = scan(body[:end])
        if not rows:
            continue
        occ = next((l for l in body if "; Occupancy:" in l), "").strip()
        pk = [r for r in rows if r[0].startswith("v_pk_") and r[0].endswith("_f32")]
        other = [r for r in rows if r not in pk]
        def mn(rs):
            same = [r[1] for r in rs if not r[2]]
            return min(same) if same else None
        print(f"{name}\n   {occ}  MFMA-result readers: {len(rows)} (packed-f32: {len(pk)}); min wait states same-block: packed {mn(pk)}, other {mn(other)}")
        for r in sorted([r for r in rows if not r[2]], key=lambda r: r[1])[:4]:
            print(f"      {r[1]:3d} wait states: {r[3]}")


if __name__ == "__main__":
    main()

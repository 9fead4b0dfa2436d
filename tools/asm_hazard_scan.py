"""Scans hipcc's assembly (-S --cuda-device-only) for the hazard found in round 6: an SGPR written by a VALU instruction
(v_readlane_b32 -- the restore of a spilled scalar -- or v_readfirstlane_b32) and read by a vector-memory instruction INSIDE an
inline-asm block fewer than 5 wait states later.  hipcc's hazard recogniser inserts the s_nop for its own instructions but does not
look into asm blocks.  Usage: python tools/asm_hazard_scan.py file.s [...]; prints every site with the distance found."""
import re, sys

VMEM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")

def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out

def wait_states(ins):
    m = re.match(r"\s*s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1

def scan(path):
    lines = open(path).read().splitlines()
    func = "?"
    hist = []          # (text, in_asm) of executed instructions in layout order, reset at labels
    in_asm = False
    found = 0
    for ln, raw in enumerate(lines, 1):
        t = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"): in_asm = True; continue
        if raw.lstrip().startswith(";;#ASMEND"): in_asm = False; continue
        if re.match(r"^_Z[\w$.]*:", raw): func = raw.split(":")[0]; hist = []; continue
        if re.match(r"^\.L\w+:", raw): hist = []; continue   # a branch target: what ran before is unknown (conservative: forget)
        if not t.strip() or t.lstrip().startswith("."): continue
        ins = t.strip()
        if in_asm and VMEM.match(ins):
            used = sregs(ins.split(None, 1)[1] if " " in ins else "")
            ws = 0
            for prev, _ in reversed(hist):
                if ws >= 5: break
                m = re.match(r"v_(readlane|readfirstlane)_b32\s+s(\d+)", prev)
                if m and int(m.group(2)) in used:
                    print("%s:%d  %s\n    in %s: s%s written by `%s` %d wait state(s) earlier" % (path, ln, ins, func[:80], m.group(2), prev, ws))
                    found += 1
                ws += wait_states(prev)
        hist.append((ins, in_asm))
        if len(hist) > 16: hist.pop(0)
    return found

if __name__ == "__main__":
    n = sum(scan(p) for p in sys.argv[1:])
    print("%d hazard site(s)" % n)
    sys.exit(1 if n else 0)

"""Guard against the round-3 miscompile: vector instructions in front of the EXEC restore of a join block.

hipcc (ROCm 7.2, gfx950) lowers `if (divergent) { .. }` to
        s_and_saveexec_b64 sN, cond ; s_cbranch_execz JOIN ; <then> ; JOIN: s_or_b64 exec, exec, sN
and everything the register allocator inserts at the top of JOIN (live-range split copies, reloads) must come AFTER that s_or_b64:
in front of it only the lanes of the `then` side are enabled (none at all in a wave that skipped it).  In the faulty build of the
one-chain SIRW streaming kernel three `v_mov_b64` split copies of values that are live in EVERY lane (the chain's cached temperature
among them) sat in front of the restore behind `if (tid == 0) shs[4] = temperature(..)`: lanes 1..255 kept stale registers, the first
half step of the doubling used hs = 0.5 eps beta = 0, and the energies were off by 75 (tools/exp_family_hmc.py).  Three independent
switches remove it (-mllvm -enable-ipra=false, -mllvm -amdgpu-spill-sgpr-to-vgpr=false, the transcendentals inlined): the copies only
appear when calls, inter-procedural register allocation and SGPR spills into VGPR lanes meet in one region.

    python -m magi_v2_amd.isa_check file.s [...]          # ISA of a --save-temps / -S device compile
(magi_v2_amd.build and magi_v2_amd.jit keep the ISA of every translation unit and run this check on it: a flagged unit is recompiled
with -mllvm -enable-ipra=false, and the build FAILS if the pattern is still there.)

For every block that is the target of an s_cbranch_execz / execnz (a join reached with a narrowed EXEC) and restores EXEC with
`s_or_b64 exec, exec, ..`, lists the instructions in front of the restore that write a vector register, touch memory or LDS under the
narrowed mask.  v_readlane / v_writelane (SGPR spill lanes: they ignore EXEC) and scalar instructions are harmless there.
Exit code 1 if any kernel has such a block."""
import re
import sys


def functions(text):
    """(name, lines) of every function body in an AMDGPU .s file."""
    out, cur, name = [], None, None
    for line in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            name, cur = m.group(1), []
            out.append((name, cur))
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith(".Lfunc_end"):
                cur = None
    return out


HARMLESS = re.compile(r"^(s_|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32|;|\.|$)")


def suspicious_blocks(lines):
    """Join blocks = targets of `s_cbranch_execz L` whose branch site saved EXEC into register X (s_and_saveexec_b64 X, .. a few lines
    above): at L everything in front of `s_or_b64 exec, exec, X` runs under the narrowed mask of the skipped region."""
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            label_at[m.group(1)] = i
    found, seen = [], set()
    for i, l in enumerate(lines):
        m = re.match(r"\s*s_cbranch_execz\s+(\.LBB\w+)", l)
        if not m or m.group(1) not in label_at:
            continue
        saved = None
        for k in range(i - 1, max(i - 8, -1), -1):
            ms = re.match(r"\s*s_and_saveexec_b64\s+(s\[\d+:\d+\]|vcc),", lines[k])
            if ms:
                saved = ms.group(1)
                break
            if re.match(r"^\.LBB\w+:", lines[k]):
                break
        if saved is None or (m.group(1), saved) in seen:
            continue
        seen.add((m.group(1), saved))
        j, before = label_at[m.group(1)] + 1, []
        while j < len(lines) and not re.match(r"^\.LBB\w+:", lines[j]):
            ins = lines[j].strip()
            if ins.startswith("s_or_b64 exec, exec, " + saved):
                if before:
                    found.append((m.group(1), before))
                break
            if re.match(r"s_(c?branch|setpc|swappc|endpgm)", ins) or re.search(r"saveexec|\bexec\b.*,|s_mov_b64 exec", ins) and not ins.startswith("v_"):
                break                     # another region begins, or EXEC is rewritten some other way: not this pattern
            if not HARMLESS.match(ins):
                before.append((j, ins))
            j += 1
    return found


def check_file(path):
    """[(function, label, [(line, instruction), ..]), ..] for one .s file (empty = clean)."""
    out = []
    for name, lines in functions(open(path).read()):
        for label, before in suspicious_blocks(lines):
            out.append((name, label, before))
    return out


def report(path, hits):
    txt = []
    for name, label, before in hits:
        txt.append(f"{path}: {name}: {label}: {len(before)} vector / memory instruction(s) in front of the EXEC restore")
        txt += [f"      +{j}: {ins}" for j, ins in before[:8]]
    return "\n".join(txt)


def main(paths):
    bad = 0
    for p in paths:
        hits = check_file(p)
        if hits:
            bad += len({h[0] for h in hits})
            print(report(p, hits))
    print("exec-prologue check:", "CLEAN" if bad == 0 else f"{bad} function(s) with instructions in front of an EXEC restore")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

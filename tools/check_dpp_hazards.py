"""Static check of the hand-written DPP instructions in a `hipcc -S` dump.

Inline assembly is opaque to the compiler's hazard recognizer, so the wait states CDNA3/4 require in
front of a DPP read are this build's responsibility (ISA guide, "manually inserted wait states"):
  * VALU writes a VGPR  -> a DPP instruction reads it as its DPP source:   2 wait states
  * VALU writes EXEC    -> any DPP instruction:                            5 wait states
Every instruction issued in between counts as one wait state, `s_nop N` as N + 1.  Labels end the
look-back conservatively only if the window is not yet satisfied (reported as "unknown").

usage: check_dpp_hazards.py <asm file> [kernel-name substring]   -> exit code 1 on a violation
"""
import re
import sys


def regs(tok):
    """'v[4:5]' -> {4,5}; 'v7' -> {7}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path, key=""):
    lines = open(path).read().split("\n")
    bad = unknown = n_dpp = 0
    name = None
    window = []          # (is_label, mnemonic, written vgprs, writes_exec, wait states it contributes)
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s or s.startswith((".", "//")):
            continue
        if s.endswith(":"):
            if s.startswith("_Z"):
                name = s[:-1]
                window = []
            else:
                window.append((True, s, set(), False, 0))
            continue
        if name is None or key not in name:
            continue
        parts = s.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if "_dpp" in mn:
            n_dpp += 1
            src = regs(ops[1]) if len(ops) > 1 else set()
            need_v, need_e, seen = 2, 5, 0
            for is_label, pmn, wr, wexec, ws in reversed(window):
                if seen >= need_e:
                    break
                if is_label:
                    unknown += 1
                    break
                if wexec and seen < need_e:
                    print(f"{name}: VALU EXEC write `{pmn}` {seen} wait states before `{s}`")
                    bad += 1
                if (wr & src) and seen < need_v:
                    print(f"{name}: `{pmn}` writes the DPP source {seen} wait states before `{s}`")
                    bad += 1
                seen += ws
        is_valu = mn.startswith("v_")
        wr = regs(ops[0]) if (is_valu and ops) else set()
        if mn.startswith("v_permlane") and len(ops) > 1 and "swap" in mn:
            wr |= regs(ops[1])
        wexec = is_valu and (mn.startswith("v_cmpx") or (ops and ops[0] == "exec"))
        ws = int(ops[0]) + 1 if mn == "s_nop" else 1
        window.append((False, mn, wr, wexec, ws))
        window = window[-12:]
    return n_dpp, bad, unknown


if __name__ == "__main__":
    n, bad, unknown = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(f"{n} DPP instructions, {bad} hazards, {unknown} look-backs cut short by a label")
    sys.exit(1 if bad else 0)

"""Check of the compiled device code for one miscompilation pattern (used by ``_lib.build`` and ``tools/isa_lint.py``).

    .LBBn_m:                        <- join block of a divergent region, entered with some or all lanes switched off
        v_mov_b32 vA, vB            <- vector instruction(s) placed BEFORE ...
        s_or_b64 exec, exec, s[..]  <- ... the instruction that switches the lanes back on

A vector register written there keeps stale contents in the lanes that skipped the region.  The register allocator of
the ROCm 7.2 compiler was seen to put a live-range-split copy in that spot (found through a wrong image in one kernel
variant: the copy fed the LDS address of one 1 KiB piece of the image copy-out); every build is scanned so that such
code does not ship unnoticed.
"""
import re

_LABEL = re.compile(r"\.LBB\d+_\d+:")
_RESTORE = re.compile(r"s_or_b64\s+exec,\s*exec,")
_OTHER_MASK = re.compile(r"s_(and|andn2|xor|or)\w*_b64\s+exec|s_mov_b64\s+exec|s_\w+saveexec")
_VECTOR = ("v_", "ds_", "global_", "buffer_", "scratch_", "flat_")
_LANE_OPS = ("v_readlane", "v_writelane", "v_readfirstlane")


def find_masked_copies(asm_path):
    """[(kernel symbol, label, [instructions])] for every occurrence of the pattern in a device assembly listing."""
    lines = open(asm_path).read().split("\n")
    kernel, hits = None, []
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
            kernel = l.split(":")[0]
        if not _LABEL.match(l.strip()):
            continue
        label, pending = l.strip().split(":")[0], []
        for j in range(i + 1, len(lines)):  # (an index, not a slice: listings run to millions of lines)
            t = lines[j].split(";")[0].strip()
            if not t:
                continue
            if _LABEL.match(t) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if _RESTORE.match(t):
                if pending:
                    hits.append((kernel, label, pending))
                break
            if _OTHER_MASK.match(t):
                break  # some other change of the lane mask: not this pattern
            op = t.split()[0]
            if op.startswith(_VECTOR) and not op.startswith(_LANE_OPS):
                pending.append(t)
    return hits

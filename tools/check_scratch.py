#!/usr/bin/env python3
"""Build-time guard for the describe kernel (called by local-features_amd/Makefile).

Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks of mkd_describe.hip and fails if an f16 (POOL = 1, or 3: the
fp6 cross-term mode, same epilogue) instantiation of mkd_pool uses scratch memory.  Why: the f16 epilogue waits for its LDS-DMA steps with COUNTED
s_waitcnt vmcnt(N); scratch spill stores count on vmcnt too and retire out of order with respect to loads, so a spill
in flight there would let a wait pass before its DMA has landed -- silently wrong descriptors, now and then.  The f32
(POOL = 2) instantiations only ever wait with vmcnt(0) and may spill.  Warnings and errors in the same file are passed
through to stderr."""
import re
import sys

text = open(sys.argv[1], errors="replace").read().splitlines()
name, bad, seen = None, [], 0
for line in text:
    if "kernel-resource-usage" not in line:
        if re.search(r"\b(warning|error|note):", line):
            print(line, file=sys.stderr)
        continue
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        continue
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
    if m and name and "mkd_pool" in name:
        t = re.search(r"mkd_poolILi(\d+)ELi(\d+)ELi(\d+)E", name)
        if t and int(t.group(2)) in (1, 3):      # F16X3 and F16_FP6: the counted waits of the shared epilogue
            seen += 1
            if int(m.group(1)):
                bad.append(f"mkd_pool<{t.group(1)}, {t.group(2)}, {t.group(3)}>: {m.group(1)} bytes/lane of scratch")
if seen == 0:
    sys.exit("check_scratch: no f16 instantiation of mkd_pool found in the remarks (was the file compiled with "
             "-Rpass-analysis=kernel-resource-usage?)")
if bad:
    sys.exit("check_scratch: the f16 describe kernel must not spill (counted vmcnt waits, see mkd_describe.hip):\n  "
             + "\n  ".join(bad))
print(f"check_scratch: {seen} f16 instantiations of mkd_pool, no scratch")

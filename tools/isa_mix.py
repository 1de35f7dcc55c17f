"""Static VALU mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): instructions of the kernel's hot loop by issue
class, weighted by loop depth 0 (straight-line count — K1's loops are rolled, so the listing's proportions are the executed ones to a few
per cent).  Full-rate class (~2.7 cycles/wave-instruction on MI355X, profiles/r1_f_valu_issue_rates.txt): plain 32-bit integer and fp32
add / mul / fma / logic / shifts / moves / cndmask; everything else VALU (packed-16, dot, sad, min/max/med3, fp64, converts, ldexp,
VOP3 three-operand integer forms, DPP / permutes excluded as they are not VALU issue) counts at ~4.4.
usage: python tools/isa_mix.py /tmp/fast.s <mangled-name-part>"""
import re, sys
src, part = open(sys.argv[1]).read().split("\n"), sys.argv[2]
FAST = re.compile(r"^v_(add|sub|subrev)_(u32|i32|co_u32|f32)\b|^v_(and|or|xor|not)_b32\b|^v_(lshrrev|lshlrev|ashrrev)_b32\b|^v_mov_b32\b|^v_(mul|fma|fmac)_f32\b|^v_cndmask_b32\b|^v_add_nc_u32\b")
inside, fast, slow, other = False, 0, 0, {}
for line in src:
    m = re.match(r"^(_Z\S+):", line)
    if m:
        inside = part in m.group(1)
        continue
    if not inside:
        continue
    ins = line.strip().split(" ")[0].split("\t")[0]
    if not ins.startswith("v_"):
        continue
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", ins)
    if FAST.match(base) and not ins.endswith("_dpp"):
        fast += 1
    else:
        slow += 1
        other[base] = other.get(base, 0) + 1
tot = fast + slow
print(f"{part}: {tot} VALU instructions in the listing, {fast} full-rate ({100 * fast / tot:.1f} %), {slow} slow-class; at 2.7 / 4.4 cycles: {(2.7 * fast + 4.4 * slow) / tot:.2f} cycles per instruction")
print("slow-class top:", sorted(other.items(), key=lambda kv: -kv[1])[:14])

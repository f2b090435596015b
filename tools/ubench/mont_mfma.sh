#!/bin/bash
# Static VALU instruction counts of the conversions an MFMA-based Montgomery reduction needs (no GPU needed), then - on a GPU -
# the MFMA / mad rates.  usage: bash tools/ubench/mont_mfma.sh [run]
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only mont_mfma.hip -o /tmp/mont_mfma.s || exit 1
python3 - <<'PY'
import re
txt = open("/tmp/mont_mfma.s").read()
def valu(name):
    body = txt.split(name + ":")[1].split("s_endpgm")[0]
    return len([l for l in body.splitlines() if re.match(r"\s+v_", l) and not re.match(r"\s+v_(mfma)", l)])
a, a0 = valu("count_limbs_to_bytes"), valu("count_baseline_9")
b, b0 = valu("count_columns_to_limbs"), valu("count_baseline_66_18")
print("limbs -> packed bytes      : %3d VALU instructions (%d with its loads/stores, %d for those alone)" % (a - a0, a, a0))
print("66 columns -> 18 limbs     : %3d VALU instructions (%d / %d)" % (b - b0, b, b0))
per_red = 2 * (a - a0) + 2 * (b - b0)
print("per reduction (m and m * p): >= %d VALU instructions of conversions, against the 108 (90 v_mad_u64_u32 + 9 v_mul_lo + 9 v_and) the"
      " mad chain spends on the same reduction - before a single operand is moved into the MFMA's cross-lane layout" % per_red)
PY
if [ "$1" = run ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 mont_mfma.hip -o /tmp/mont_mfma && /tmp/mont_mfma; fi

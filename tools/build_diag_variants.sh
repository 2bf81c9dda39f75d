#!/bin/bash
# builds tools/attn_asm_diag_<name> for a list of schedule-knob settings (ASM_* environment of gen/attn_asm_gen.py)
set -e
cd "$(dirname "$0")/.."
i=0
while read -r name knobs; do
  [ -z "$name" ] && continue
  case "$knobs" in *NO_*) ;; *) env $knobs python3 tools/emu_check.py > /dev/null 2>&1 || { echo "EMULATOR CHECK FAILED: $name ($knobs) -- not built"; continue; } ;; esac
  env $knobs python3 longlive_amd/csrc/gen/attn_asm_gen.py --diag tools/attn_asm_body_d.inc > /dev/null 2>&1 || { echo "gen failed: $name"; exit 1; }
  hipcc --offload-arch=gfx950 -O2 -Iinclude -Ilonglive_amd/csrc tools/attn_asm_diag.hip -o tools/attn_asm_diag_$name 2>&1 | grep -E "error" && exit 1
  echo "built $name ($knobs)"
done

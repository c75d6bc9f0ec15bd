#!/bin/bash
# usage: scripts/kernel_resources.sh [build/*.o ...]  -- VGPRs / SGPRs / spills / scratch / LDS of the kernels in the given
# objects (default: every unit under build/); no GPU needed
OBJS=${@:-build/*.o}
TMP=$(mktemp -d)
for o in $OBJS; do
  objcopy -O binary --only-section=.hip_fatbin $o $TMP/fat.bin 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/k.co --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/k.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    def g(k):
        r = re.search(r'\.' + k + r':\s+(\d+)', blk)
        return int(r.group(1)) if r else -1
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    try: name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except Exception: pass
    m = re.match(r'(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)', name)
    print('%-44s vgpr %3d sgpr %3d spill_v %4d spill_s %4d scratch %5d lds %6d' % (m.group(1) if m else name[:44], g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
done
rm -rf $TMP

"""ISA audit for the kernels that read LDS through inline asm (mma_f64.h: ds_read64 + lgkm_wait + asm MFMAs): hipcc does not know that the
destination of an asm ds_read is in flight until our own s_waitcnt, so under register pressure it may copy or spill it early
(cdna_hip_programming.md section 5.7, item 1).  This script walks the -save-temps assembly and reports every compiler-generated
instruction that touches such a register between the read and the wait that covers it.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c gparml_amd/csrc/X.hip -save-temps=obj -o /tmp/X.o; python3 tools/audit_asm_reads.py X-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re, sys
# audit: between an inline-asm ds_read (VGPR destination) and the next inline-asm s_waitcnt lgkmcnt that covers it, no compiler-generated
# instruction may touch the destination registers
def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
def audit(path):
    kern = None; in_asm = False; pending = {}   # reg -> line of the read
    bad = 0; reads = 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(':') and s.startswith('_Z'): kern = s[:-1]; pending = {}
        if s.startswith(';;#ASMSTART'): in_asm = True; continue
        if s.startswith(';;#ASMEND'): in_asm = False; continue
        if not s or s.startswith(';') or s.startswith('.'): continue
        toks = re.split(r'[ ,\t]+', s)
        if in_asm:
            if toks[0].startswith('ds_read'):
                for r in regs(toks[1]): pending[r] = ln
                reads += 1
            elif toks[0] == 's_waitcnt' and 'lgkmcnt' in s:
                n = int(re.search(r'lgkmcnt\((\d+)\)', s).group(1))
                # the oldest reads are complete: keep only the newest n reads' registers (2 regs per read)
                keep = sorted(set(pending.values()))[-n:] if n > 0 else []
                pending = {r: l for r, l in pending.items() if l in keep}
            continue
        # compiler instruction: does it touch a pending register?
        used = set()
        for t in toks[1:]: used |= regs(t.rstrip(','))
        hit = used & set(pending)
        if hit:
            bad += 1
            if bad <= 8: print('%s line %d: %s   touches v%s loaded at line %s' % (kern[:50] if kern else '?', ln, s, sorted(hit), sorted({pending[h] for h in hit})))
    print(path, 'asm ds_reads:', reads, 'suspicious compiler accesses:', bad)
for p in sys.argv[1:]: audit(p)

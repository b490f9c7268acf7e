"""ISA audit for inline-asm MFMAs (mma_f64.h: mfma444_acc / mfma444_zero and the int8 MFMAs of p1i8.hip): hipcc does not model the instruction
inside an asm string, so it inserts none of the wait states the hazard recogniser would.  Two patterns produced wrong numbers in this repository:
  (a) r03, psi2_tile.hip: an accumulator zeroed in C++ is rematerialised as v_mov directly in front of its first asm MFMA (VALU write -> MFMA read
      of SrcC without wait states);
  (b) r05, gemm32_tile inside gs_tail128_kernel: such a v_mov landed on the register the PREVIOUS MFMA was still reading as its B operand
      (write-after-read inside the MFMA's multi-pass operand read): columns 4..7 of every 32 x 32 tile wrong, run-to-run different.
(Pattern (b) is (a) in disguise: the register had just been the B operand of the previous MFMA, the zeroing move was put right behind that MFMA and
right in front of the one that reads it as SrcC.  A VALU write of a register that an in-flight MFMA read as SrcA / SrcB is NOT a hazard -- those are read
at issue; production kernels contain such writes and are bit-exact against the oracle -- so only SrcC is tracked behind an MFMA.)
This script walks the -save-temps assembly and reports every compiler-generated (non-asm) VALU instruction that, within WINDOW
cycles behind an asm MFMA's issue, writes the VGPRs that MFMA reads as SrcC -- VALU instructions only: the data of an LDS or global load
arrives long after any MFMA in flight has read its operands --, and every compiler-generated VALU write of an MFMA's source directly in front of it.
Issue cycles are counted as 16 per MFMA (4 passes) and 4 per other instruction.  usage: tools/build_lib.sh --asm; python3 tools/audit_mfma_hazards.py /tmp/asm/*-gfx950.s
r06, --ab: ALSO report every compiler-generated VALU write of a register that an asm MFMA issued within the last AB_WINDOW cycles reads as SrcA / SrcB (the
round-5 review's question about psi2_tile.hip / psi2.hip).  The ISA reads A and B at issue and LLVM's hazard recogniser knows no such hazard (only SrcC
is read in later passes), so these are reported as 'note', counted separately and do not fail the audit."""
import re
import sys

RESULT_WINDOW = 17   # cycles behind an MFMA's issue in which its result must not be read by a non-MFMA instruction (4 passes + write-back)
WINDOW = 16      # cycles behind the MFMA's issue in which it may still read its operands (4 passes of 4 cycles)
AB_WINDOW = 16
CHECK_AB = False


def regs(tok):
    tok = tok.rstrip(',')
    m = re.match(r'[va]\[(\d+):(\d+)\]$', tok)
    if m:
        return {(tok[0], r) for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r'[va](\d+)$', tok)
    return {(tok[0], int(m.group(1)))} if m else set()


def audit(path):
    kern, in_asm = None, False
    recent = []          # (age, line, text, source registers) of the last asm MFMAs
    recent_ab = []       # the same for the A / B operands (--ab)
    notes = 0
    results = []         # (age, line, text, destination registers) of the last asm MFMAs: a compiler-generated READ inside the latency window is hazard (c)
    bad = mfmas = 0
    prev_compiler_write = None     # (line, text, dst regs) of the previous instruction if compiler-generated
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(':') and s.startswith('_Z'):
            kern, recent, results, prev_compiler_write = s[:-1], [], [], None
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        toks = re.split(r'[ \t]+', s.split(';')[0].strip())
        op = toks[0]
        cost = 16 if op.startswith('v_mfma') else 4
        if op == 's_nop':
            try:
                cost = int(toks[1]) + 1
            except (IndexError, ValueError):
                pass
        if in_asm and op.startswith('v_mfma'):
            mfmas += 1
            srcs = set()
            for t in toks[2:5]:
                srcs |= regs(t)
            srcc = regs(toks[4]) if len(toks) > 4 else set()
            # (a) the instruction right in front is a compiler-generated write of one of this MFMA's sources
            if prev_compiler_write and (prev_compiler_write[2] & srcs):
                bad += 1
                if bad <= 10:
                    print('%s line %d: "%s" reads %s written by the compiler one instruction earlier: "%s"' % (
                        (kern or '?')[:60], ln, s, sorted(prev_compiler_write[2] & srcs), prev_compiler_write[1]))
            recent = [(a + cost, l, t, r) for (a, l, t, r) in recent if a + cost < WINDOW]
            recent.append((0, ln, s, srcc))
            recent_ab = [(a + cost, l, t, r) for (a, l, t, r) in recent_ab if a + cost < AB_WINDOW]
            recent_ab.append((0, ln, s, (regs(toks[2]) | regs(toks[3])) - srcc))
            results = [(a + cost, l, t, r) for (a, l, t, r) in results if a + cost < RESULT_WINDOW]
            results.append((0, ln, s, regs(toks[1])))
            prev_compiler_write = None
            continue
        dst = regs(toks[1]) if len(toks) > 1 and op.startswith('v_') and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_mfma')) else set()
        if not in_asm and op.startswith(('v_', 'global_store', 'buffer_store', 'flat_store', 'ds_write')) and not op.startswith('v_mfma'):
            # (c) a compiler-generated instruction READS an asm MFMA's result inside its latency window (the source code puts mfma_drain() = s_nop 15
            # in front of such reads; a register copy the compiler inserts on its own -- e.g. to merge two branches -- has no such protection)
            srcs_here = set()
            for t in toks[2:] if op.startswith('v_') else toks[1:]:
                srcs_here |= regs(t)
            for (a, l, t, r) in results:
                if srcs_here & r:
                    bad += 1
                    if bad <= 10:
                        print('%s line %d: "%s" reads %s, the result of the asm MFMA issued %d cycle(s) earlier (line %d): "%s"' % (
                            (kern or '?')[:60], ln, s, sorted(srcs_here & r), a, l, t))
        if not in_asm and dst:
            for (a, l, t, r) in recent:
                if dst & r:
                    bad += 1
                    if bad <= 10:
                        print('%s line %d: "%s" overwrites %s, an operand of the asm MFMA issued %d cycle(s) earlier (line %d): "%s"' % (
                            (kern or '?')[:60], ln, s, sorted(dst & r), a, l, t))
            if CHECK_AB:
                for (a, l, t, r) in recent_ab:
                    if dst & r:
                        notes += 1
                        if notes <= 6:
                            print('note: %s line %d: "%s" writes %s, an A/B operand of the asm MFMA issued %d cycle(s) earlier (line %d)' % ((kern or '?')[:50], ln, s, sorted(dst & r), a, l))
            prev_compiler_write = (ln, s, dst)
        else:
            prev_compiler_write = None
        recent = [(a + cost, l, t, r) for (a, l, t, r) in recent if a + cost < WINDOW]
        recent_ab = [(a + cost, l, t, r) for (a, l, t, r) in recent_ab if a + cost < AB_WINDOW]
        results = [(a + cost, l, t, r) for (a, l, t, r) in results if a + cost < RESULT_WINDOW]
    print('%s: asm MFMAs %d, hazards %d%s' % (path, mfmas, bad, (', A/B operand overwrites inside the pass window (no hazard by the ISA): %d' % notes) if CHECK_AB else ''))
    return bad


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--ab']
    CHECK_AB = '--ab' in sys.argv[1:]
    sys.exit(1 if sum(audit(p) for p in args) else 0)

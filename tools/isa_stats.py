"""Instruction statistics of one kernel in a hipcc -save-temps .s file (tools for reading the hot loops).
usage: isa_stats.py FILE.s SUBSTRING [--loop]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\S+):', s, re.M) if pat in n]
for n in names:
    i = s.index(n + ':')
    j = s.index('.Lfunc_end', i)
    body = s[i:j]
    print(n[-60:], 'lines', body.count('\n'), 'mfma', body.count('v_mfma'), 'lds-dma', len(re.findall(r'buffer_load_dwordx4.*lds', body)),
          'ds_write_b128', body.count('ds_write_b128'), 'ds_write_b64', body.count('ds_write_b64'), 'ds_read_b128', body.count('ds_read_b128'))
    print('   vmcnt waits:', re.findall(r's_waitcnt vmcnt\((\d+)\)', body))
    k = s.index('.amdhsa_kernel ' + n)
    blk = s[k:k + 4000]
    print('  ', re.findall(r'\.amdhsa_next_free_vgpr \d+|\.amdhsa_accum_offset \d+|\.amdhsa_group_segment_fixed_size \d+', blk),
          re.findall(r'; ScratchSize: \d+', s[j:j + 3000]))
    if '--loop' in sys.argv:
        # the innermost loop that contains MFMAs: print it
        labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(\.LBB\d+_\d+):', body, re.M)]
        best = None
        for a, lab in labels:
            for m in re.finditer(r's_cbranch_\w+ ' + re.escape(lab) + r'\b', body):
                if m.start() > a and 'v_mfma' in body[a:m.start()]:
                    if best is None or m.start() - a < best[1] - best[0]:
                        best = (a, m.end())
        if best:
            print(body[best[0]:best[1]])

"""Instruction / register statistics of the production kernel instantiations from csrc/_build/render_fast.s (make asm)."""
import os, re, sys
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "heightmap-ray-marcher_amd", "csrc", "_build", "render_fast.s")
s = open(path).read()
parts = re.split(r'\t\.type\t(_ZN4hmrm13k_render_fastI[^,]+),@function\n', s)
want = sys.argv[1:] or ["ILi2ELb0ELi0ELi1ELi0E", "ILi1ELb0ELi0ELi1ELi0E", "ILi3ELb0ELi0ELi1ELi0E"]
for i in range(1, len(parts), 2):
    if not any(w in parts[i] for w in want):
        continue
    f = parts[i + 1]
    code = f.split('s_endpgm')[0]
    lines = [l.strip() for l in code.split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    g = lambda pat: re.search(pat, f).group(1)
    print(re.search(r'k_render_fastI(\w+?)EvNS', parts[i]).group(1), 'instrs', len(lines), 'valu', sum(l.startswith('v_') for l in lines),
          'f64', sum('_f64' in l for l in lines), 'salu', sum(l.startswith('s_') for l in lines),
          'branches', sum(l.startswith('s_cbranch') for l in lines), 'vgpr', g(r'; NumVgprs: (\d+)'), 'occupancy', g(r'; Occupancy: (\d+)'))

"""Basic-block view of one k_render_fast instantiation (csrc/_build/render_fast.s, `make asm`): instructions,
VALU, fp64 and memory instructions per block and where each block branches.  The loop blocks are the ones
whose cost repeats per trip; tools/isa_stats.py has the totals.
usage: python tools/isa_blocks.py [instantiation-substring, default the C3 production kernel] [--dump]"""
import os, re, sys
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "heightmap-ray-marcher_amd", "csrc", "_build", "render_fast.s")
s = open(path).read()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[0] if args else "ILi2ELb0ELi0ELi1ELi0E"
parts = re.split(r'\t\.type\t(_ZN4hmrm13k_render_fastI[^,]+),@function\n', s)
body = None
for i in range(1, len(parts), 2):
    if want in parts[i]:
        body = parts[i + 1].split('s_endpgm')[0]
        break
if body is None:
    sys.exit("no such instantiation")
if "--dump" in sys.argv:
    print(body)
    sys.exit(0)
blocks, cur = [], ['entry', []]
for l in body.split('\n'):
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    if t.startswith('.'):
        continue
    cur[1].append(t.split(';')[0].strip())
blocks.append(cur)
tot = 0
for name, ins in blocks:
    v = sum(x.startswith('v_') for x in ins)
    br = [x for x in ins if x.startswith(('s_cbranch', 's_branch'))]
    print(f"{name:12s} n={len(ins):4d} valu={v:4d} f64={sum('_f64' in x for x in ins):3d} "
          f"mem={sum(x.startswith(('global_', 'buffer_', 's_load', 'ds_', 'flat_')) for x in ins):2d}  ->",
          ' '.join(b.split()[-1] + ('?' if 'cbranch' in b else '') for b in br))
    tot += v
print("valu", tot)

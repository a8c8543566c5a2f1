import re,sys
lines=[l for l in open(sys.argv[1]) if l.strip().startswith('[')]
ev=[(float(re.search(r'\[\s*([\d.]+) ms\]',l).group(1)), l.split(']')[1].strip()) for l in lines]
idx=[i for i,(t,s) in enumerate(ev) if s.startswith('shard 0: front (')]
start=idx[2]
t0=ev[start][0]
for t,s in ev[start:]:
    if 'front (' in s: f0=t; name=s
    elif 'front done' in s: print(f"{f0-t0:8.1f} {t-f0:6.1f} ms  {name}")
    elif 'coder launched' in s: print(f"{t-t0:8.1f}        coder launched")
    elif 'event reached' in s: print(f"{t-t0:8.1f}        {s}")

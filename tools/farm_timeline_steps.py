import sys, collections
runs=[]; cur=None
for line in open(sys.argv[1]):
    if line.startswith('# run'):
        cur=[]; runs.append((line.strip(),cur))
    elif line.startswith('step'):
        p=line.split(); cur.append((int(p[1]),int(p[2]),float(p[3]),float(p[5])))
hdr,steps=runs[-1]   # the timed run of the resident leg is the one with most steps before host-fed... take the run with 40 steps
for h,s in runs:
    if '40 steps' in h: hdr,steps=h,s; break
print(hdr)
by=collections.defaultdict(list)
for g,s,t0,t1 in steps: by[s].append((t0,t1))
prev_end=0
for s in sorted(by):
    v=by[s]; dur=[b-a for a,b in v]
    print("step %2d: mean group-step %.2f ms  min %.2f max %.2f   last finish at %.2f ms"%(s, 1e3*sum(dur)/len(dur), 1e3*min(dur), 1e3*max(dur), 1e3*max(b for a,b in v)))

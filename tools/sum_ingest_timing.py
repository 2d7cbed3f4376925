import re,collections,sys
acc=collections.defaultdict(list)
for l in open(sys.argv[1]):
    m=re.match(r"\[ingest\] (\S+) (.+?) ([0-9.]+) ms", l)
    if m: acc[m.group(1)+" "+m.group(2)].append(float(m.group(3)))
for k,v in acc.items(): print(k.ljust(40), "n",len(v), "mean %.1f  min %.1f  max %.1f"%(sum(v)/len(v), min(v), max(v)))

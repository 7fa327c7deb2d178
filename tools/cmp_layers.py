import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
kb={tuple(r['B,H,W,Cin,Cout,k,splitk']) if 'B,H,W,Cin,Cout,k,splitk' in r else None:r for r in b}
for r in a:
    if 'B,H,W,Cin,Cout,k,splitk' not in r or r['class']!='conv3x3_winograd': continue
    k=tuple(r['B,H,W,Cin,Cout,k,splitk']); o=kb.get(k)
    if o: print(k, r['launches_per_step'], f"{r['ms_per_launch']*1e3:7.1f} {o['ms_per_launch']*1e3:7.1f} {r['ms_per_launch']/o['ms_per_launch']-1:+.1%}")

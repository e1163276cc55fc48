import sys; sys.path.insert(0,'.')
from larynx_amd.engine import Engine
eng=Engine(0)
for (B,Cin,Cout,K,dil,L) in [(1,128,128,11,1,39936),(1,1024,128,11,1,39936),(1,128,128,3,1,39936),(1,1024,128,3,1,39936),(1,128,128,7,3,39936),(4,128,128,11,1,39936),(1,256,256,11,1,4992),(1,2048,256,11,1,4992)]:
    ms=eng.bench_conv1d(B,Cin,Cout,K,dil,L,-1,30)
    fl=2.0*B*Cin*Cout*K*L
    print(f"B{B} Cin{Cin} Cout{Cout} K{K} L{L}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")

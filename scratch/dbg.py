import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, chainspecs as CS
import vr180_convert_amd as V
from oracle import oracle as O
g=np.load('tests/golden/maps_small.npz')
name='equirect_decoder_lat_x'
spec,out,inp,radius=CS.SMALL_CASES[name]
xm,ym=V.get_map(CS.to_product(spec),radius=radius,size_input=inp,size_output=out)
gx,gy=g[name+'__x'],g[name+'__y']
ox,oy=O.get_map(spec,radius=radius,size_input=inp,size_output=out,f64=True)
for a,gg,o,nm in ((xm,gx,ox,'x'),(ym,gy,oy,'y')):
    bad=np.argwhere(CS.buckets(a)!=CS.buckets(gg))
    print(nm,'mismatches',len(bad))
    for (j,i) in bad[:12]:
        print('  px',j,i,'gpu',repr(a[j,i]),'gold',repr(gg[j,i]),'oracle f64',repr(o[j,i]), 'x32', o[j,i]*32)

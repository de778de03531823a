import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge
po=ge.load_oracle(); ge.load()
from map_merge_amd import synth
N=int(sys.argv[1])
t=time.time(); world,maps=synth.synth_maps(16,N); print('gen 16 maps',time.time()-t, 'extent', world.extent, 'patches', len(world.patches['o']))
x,c,T=maps[0]; p=synth.pack_points(x,c)
t=time.time(); d=po.downsample(p,0.1); print('down',len(d),time.time()-t)
t=time.time(); o=po.remove_outliers(d,0.8,50); print('outl',len(o),time.time()-t)
t=time.time(); n=po.normals(o,0.6); print('nrm',time.time()-t)
t=time.time(); kp,sc=po.keypoints_sift(o,0.1,3,3,5.0); print('sift',len(kp),time.time()-t)
t=time.time(); kp2,desc=po.descriptors_fpfh(o,n,kp,0.8); print('fpfh',len(kp2),time.time()-t)

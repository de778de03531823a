import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge
po=ge.load_oracle(); mm=ge.load()
from map_merge_amd import synth
world,maps=synth.synth_maps(2,12000,overlap_step=0.35)
F=[]
for x,c,T in maps:
    raw=synth.pack_points(x,c); down=po.downsample(raw,0.1); filt=po.remove_outliers(down,0.8,50); nrm=po.normals(filt,0.6)
    kp,_=po.keypoints_sift(filt,0.1,3,3,5.0); kp,desc=po.descriptors_fpfh(filt,nrm,kp,0.8); F.append(desc)
ctx=mm.Context(0)
da,db=ctx.descriptors(F[0]),ctx.descriptors(F[1])
for k in (1,5,10):
    got=ctx.findFeatureCorrespondences(da,db,k); ref=po.find_correspondences(F[0],F[1],k)
    print(k,len(got),len(ref))
    if len(got)==len(ref):
        bad=np.where((got['index_match']!=ref['index_match'])|(got['distance'].view(np.uint32)!=ref['distance'].view(np.uint32)))[0]
        print(' mismatches',len(bad))
        for i in bad[:5]: print('  ',got[i],ref[i])
idx,d2=po.desc_knn(F[0],F[1],10)
# brute force numpy check in float32 sequential
a=F[0][0]; r=np.zeros(len(F[1]),np.float32)
for d in range(33):
    df=(a[d]-F[1][:,d]).astype(np.float32); r=(r+df*df).astype(np.float32)
o=np.argsort(r,kind='stable')[:10]; print(o, idx[0]); print(r[o].view(np.uint32)==d2[0].view(np.uint32))

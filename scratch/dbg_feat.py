import importlib, sys, numpy as np
sys.path.insert(0,'.')
from oracle import pyoracle as O
ife=importlib.import_module("image-feature-extraction_amd")
synth=importlib.import_module("image-feature-extraction_amd.synthetic")
O.set_threads(16)
shape=(64,64,64)
img=synth.volume_f32(shape, synth.SEED_CONFIG[3])
mask=np.minimum(synth.mask_ellipsoids(shape),1).astype(np.uint8); mask[0,0,:]=1
ctx=ife.Context(0)
np.set_printoptions(precision=9, linewidth=220)
for sigma in (1.0,2.0):
    got=ctx.emphysema_features(img,mask,[sigma])[0]
    ref=O.emphysema_features(img,mask,sigma)
    S=ref[...,0]
    # hessian from oracle on unmasked S: recompute S unmasked
    Sfull=O.normalized_gaussian_convolution(img, mask.astype(np.float32), sigma)
    H=O.hessian3d(Sfull)
    bad=np.argwhere((got!=ref).any(-1))
    print('sigma',sigma,'mismatching voxels',len(bad))
    for b in bad[:6]:
        z,y,x=b
        A=H[z,y,x]
        print(' vox',b,'H',A)
        print('   got',got[z,y,x,2:]); print('   ref',ref[z,y,x,2:])
        print('   gpu batch fast', ctx.eigenvalues(A[None])[0], ' oracle', O.eig3(A[None])[0], O.eig3(A[None].astype(np.float64))[0])

import importlib, numpy as np, sys
sys.path.insert(0,'/root/repo')
pkg=importlib.import_module('image-feature-extraction_amd'); synth=importlib.import_module('image-feature-extraction_amd.synthetic')
for shape in [(40,44,48),(37,50,23),(100,30,70)]:
    img=synth.volume_f32(shape,5); m=np.minimum(synth.mask_ellipsoids(shape),1).astype(np.uint8)
    with pkg.Context(0) as c:
        a=c.emphysema_features(img,m,[1.0,3.0],(0.7,0.8,1.3))
        c.set_option(pkg.OPT_IIR_BLOCK,12)
        b=c.emphysema_features(img,m,[1.0,3.0],(0.7,0.8,1.3))
    print(shape, np.array_equal(a.view(np.uint32),b.view(np.uint32)))

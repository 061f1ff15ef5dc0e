import sys, os, importlib, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
t=time.time()
gm = importlib.import_module('gnark-crypto_amd')
L = gm._lib.load()
import torch
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.device_count(), time.time()-t)
print("gmsm devices", L.gmsm_device_count())
import oracle
o = oracle.Oracle('bn254','g1'); g = gm.G1Affine('bn254')
pts = o.gen_points(1000, 5, 7); rng = np.random.default_rng(1)
sc = rng.integers(0, 2**62, size=(1000,4), dtype=np.uint64)
aff, err = g.MultiExp(pts, sc); print(err, (aff == o.msm_affine(pts, sc)).all())
x = torch.from_numpy(pts.view(np.int64)).cuda(); y = torch.from_numpy(sc.view(np.int64)).cuda()
j = g.multiexp_device(x.data_ptr(), y.data_ptr(), 1000, torch.cuda.current_stream().cuda_stream)
print((g.jac_to_affine(j) == aff).all())

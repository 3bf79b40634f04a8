import numpy as np, time, ctypes as C
from nmfgpu_amd._lib import library
lib=library()
rs=np.random.RandomState(0)
V=np.asfortranarray(rs.random_sample((2000,5000)).astype(np.float32))
Cl=np.asfortranarray(np.zeros((2000,64),np.float32))
memb=np.zeros(5000,np.uint32); it=C.c_uint(0)
t=time.time(); st=lib.nmfamd_host_kmeans_f32(C.c_void_p(V.ctypes.data),C.c_long(2000),2000,5000,C.c_void_p(Cl.ctypes.data),C.c_long(2000),64,C.c_void_p(memb.ctypes.data),1,10,C.c_double(0.0),C.byref(it)); print(time.time()-t, st, it.value, np.bincount(memb,minlength=64)[:8])

import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(2,512,4).astype(np.int64)
for k,name in enumerate(("H tail (W^T V)","W tail (V H^T)")):
    s=a[k]; live=s[:,0]>0; s=s[live]; t0=s[:,0].min()
    print(name, "blocks", live.sum())
    ent=(s[:,0]-t0)/100.; arr=(s[:,1]-t0)/100.; wd=np.where(s[:,2]>0,(s[:,2]-t0)/100.,np.nan); end=np.where(s[:,3]>0,(s[:,3]-t0)/100.,np.nan)
    idx=np.nonzero(live)[0]
    print("  entry  us: min %.1f max %.1f; >1us late: %d blocks: %s"%(ent.min(),ent.max(),(ent>1).sum(), idx[ent>1][:40]))
    print("  arrive us: min %.1f median %.1f max %.1f (block %d)"%(arr.min(),np.median(arr),arr.max(), idx[arr.argmax()]))
    print("  latest arrivals:", [(int(idx[i]), round(float(ent[i]),1), round(float(arr[i]),1)) for i in np.argsort(arr)[-12:]])
    print("  wait done: min %.1f max %.1f ; end: max %.1f"%(np.nanmin(wd),np.nanmax(wd),np.nanmax(end)))

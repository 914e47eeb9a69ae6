import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,4).astype(np.int64)
a=a[(a>0).all(axis=1)]
pro=a[:,1]-a[:,0]; main=a[:,2]-a[:,1]; epi=a[:,3]-a[:,2]; tot=a[:,3]-a[:,0]
span=a[:,3].max()-a[:,0].min()
print("tiles",len(a),"ticks: prologue med %d  mainloop med %d  epilogue med %d  total med %d ; kernel span %d"%(np.median(pro),np.median(main),np.median(epi),np.median(tot),span))
print("shares: pro %.1f%% main %.1f%% epi %.1f%%"%(100*pro.sum()/tot.sum(),100*main.sum()/tot.sum(),100*epi.sum()/tot.sum()))
print("sum(tile lifetimes)/span = %.1f resident tiles on average" % (tot.sum()/span))

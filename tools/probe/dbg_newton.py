import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import tests.test_gpu_newton as T
for (N,nx,ex) in [(6,96,'lds'),(6,96,'global'),(6,96,'lds')]:
    os.environ['CATINT_NEWTON_EXCHANGE']=ex; print(ex)
    for maxit in (50,50,50):
        got,ref=T.run_both(N,nx,B=5,seed=N*1000+nx,newton_kw=dict(maxit=maxit))
        c,phi,its,st=got; rc,rphi,rit=ref
        d=np.abs(c-rc)/np.abs(rc).max(axis=2,keepdims=True)
        print(N,nx,'maxit',maxit,'its',its,rit,'max rel diff per lane',d.max(axis=(1,2)),'phi',np.abs(phi-rphi).max(axis=1))
        b=int(np.argmax(d.max(axis=(1,2)))); k=int(np.argmax(d[b].max(axis=1)))
        print('   worst lane',b,'species',k,'at i',int(np.argmax(d[b,k])), d[b,k][:6], d[b,k][-4:])

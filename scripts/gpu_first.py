import sys, time, numpy as np
sys.path.insert(0, '.')
import lanczos_amd as la, oracle
def rel(a,b): return np.max(np.abs(a-b)/np.maximum(np.abs(b),1e-300))
for (d,N,steps) in ((1,10000,30),(3,24,30),(2,100,30)):
    A=la.synth_laplacian(d,N); n=A.dimension
    v0=la.gaussian_fill(123456789,0,n)
    x=np.random.default_rng(1).standard_normal(n)
    yo=oracle.matvec_crs(A.row_head_indexes,A.column_index,A.values,x)
    for spmv in ("vector","stream","pipe","pipe_nt"):
        for reorth in ("full","mgs","full2","none","cuda_compat"):
            with la.Engine(A,steps+1,reorth=reorth,spmv=spmv) as e:
                y=e.spmv(x)
                e.set_start(v0)
                al,be=e.run(steps)
                o=oracle.lanczos_crs(A.row_head_indexes,A.column_index,A.values,v0,4,steps+1,0.0,
                                     reorth={"full":"full","mgs":"full","full2":"full","none":"none","cuda_compat":"cuda_compat"}[reorth])
                r=la.ritz_values(al,be,steps,4)
                print(d,N,e.spmv_name,reorth,"spmv err",np.max(np.abs(y-yo)),"alpha rel",rel(al,o.alpha),"beta rel",rel(be,o.beta),"ritz rel",rel(r,o.eigenvalues),flush=True)

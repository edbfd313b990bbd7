import sys, torch, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
import cases, harness
from druggen_amd import functional as dgf
import test_hip_bf16 as T
for name in ["tiny_relu","c1_b4","c2_b2","c5_b2","chembl_b4"]:
    case=cases.CASES[name]; fx=harness.load_fixture(name)
    for mode in (torch.float32, torch.bfloat16):
        cfg,G,D=T._build(case)
        inp=harness.torch_inputs(case, torch.float32, "cuda")
        with dgf.activations(mode):
            res=harness.run_step(G,D,T._d_loss,T._g_loss,inp,case["lambda_gp"])
        out=[]
        for grp in ("D.grad","G.grad"):
            try:
                harness.compare_grad_table(case,fx,"ref64",grp,res[grp],1e9)
            except Exception as e:
                out.append(str(e)[:80]); continue
            # collect all rel errs
            errs=[]
            table=res[grp]
            names=[k for k in table if table[k] is not None]
            import json
            wants={k:fx[f"ref64/{grp}/{k}"] for k in names}
            full=case["full"]
            tot=np.sqrt(sum(float((w**2).sum()) if full else float(w[0]**2) for w in wants.values()))
            floor=tot/np.sqrt(len(wants))
            for idx,k in enumerate(table.keys()):
                if table[k] is None: continue
                g=table[k].detach().double().cpu().numpy()
                if full:
                    err=np.linalg.norm((g-wants[k]).ravel()); sc=max(np.linalg.norm(wants[k].ravel()),floor)
                else:
                    gs=cases.summarise(g, idx); err=np.abs(gs-wants[k]).max(); sc=max(wants[k][0],floor)
                errs.append((err/sc,k))
            errs.sort(reverse=True)
            out.append((grp, [(f"{e:.3f}",k) for e,k in errs[:4]], f"median {np.median([e for e,_ in errs]):.4f}"))
        dl=abs(float(res["d_loss"])-float(fx["ref64/d_loss"]))/max(1,abs(float(fx["ref64/d_loss"])))
        gl=abs(float(res["g_loss"])-float(fx["ref64/g_loss"]))/max(1,abs(float(fx["ref64/g_loss"])))
        print(name, mode, f"d_loss err {dl:.2e} g_loss err {gl:.2e}")
        for o in out: print("   ",o)

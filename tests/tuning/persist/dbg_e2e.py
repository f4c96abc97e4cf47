import json, os, subprocess, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
os.environ["MVAE_TUNING"]="1"
import molecular_vae_amd as mv
from molecular_vae_amd import data as D, ops
dev=torch.device("cuda",0)
out="/tmp/e2e_dbg"; os.makedirs(out, exist_ok=True)
rep=os.path.join(out,"report.json")
subprocess.run([sys.executable, os.path.join(ROOT,"examples","train_zinc.py"),"-b","128","--epochs","2","--n_synth","3072","--out_dir",out,"--report",rep],check=True,capture_output=True)
r=json.load(open(rep)); print("child val per epoch", [e["val_loss"] for e in r["epochs"]])
ck=torch.load(r["checkpoint"],map_location="cpu",weights_only=False)
smiles=D.synthetic_smiles(3072,seed=0); vocab=D.build_vocab(smiles,120); enc=D.encode_smiles(smiles,vocab,120)
msk=np.random.RandomState(1).rand(len(enc))<0.8
test_ds=D.DeviceDataset(enc[~msk],len(vocab),dev)
for persist in ("1","0"):
    ops.PERSIST_DEFAULT=persist
    model=mv.MolecularVAE(i=ck["max_len"],c=len(ck["charset"]),o=ck["latent_size"]).to(dev)
    opt=mv.FusedAdam(model.parameters(),lr=1.0,max_grad_norm=3.0)
    mv.load_checkpoint(r["checkpoint"],model,opt)
    for k in range(3):
        val,acc=mv.evaluate(model,mv.make_loss_function(120),test_ds.batches(128,shuffle=False,drop_last=False))
        print("persist",persist,"eval",k,val,acc)

import sys, torch, torch.nn.functional as F
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from helpers import make_case, hip_model_like, rel_err
from oracle.pyg_restatement import OracleGNN
dev=torch.device('cuda:0')
for case in [((32,),8,8,1,'GRAND'), ((11,11),2,8,4,'GRAND_plus'), ((32,32),2,64,4,'GRAND_plus')]:
    opt, ds, data, oracle = make_case(*case)
    model = hip_model_like(oracle, ds, opt, dev)
    tgt = data.x_phys if data.x_phys.dim()==2 else data.x_phys.unsqueeze(-1)
    ref = oracle(data); F.mse_loss(ref,tgt).backward()
    o64 = OracleGNN(ds, dict(opt)).double(); o64.load_state_dict({k:v.double() for k,v in oracle.state_dict().items()})
    d64 = data.clone()
    for k in ('x_comp','f_tensor','uu_tensor'): setattr(d64,k,getattr(d64,k).double())
    r64 = o64(d64); F.mse_loss(r64,tgt.double()).backward()
    out = model(data.clone().to(dev)); F.mse_loss(out,tgt.to(dev)).backward(); torch.cuda.synchronize()
    print(case, 'x: hip-vs-32', rel_err(out,ref)[0], 'hip-vs-64', rel_err(out,r64)[0], 'o32-vs-64', rel_err(ref,r64)[0])
    for n in ('lin_query.weight','lin_query.bias','lin_key.weight'):
        g32=dict(oracle.conv_layers[0].named_parameters())[n].grad; g64=dict(o64.conv_layers[0].named_parameters())[n].grad; gh=dict(model.conv_layers[0].named_parameters())[n].grad
        print('   ',n,'hip-vs-32 %.2e hip-vs-64 %.2e o32-vs-64 %.2e |g|max %.2e'%(rel_err(gh,g32)[0],rel_err(gh,g64)[0],rel_err(g32,g64)[0], g64.abs().max()))

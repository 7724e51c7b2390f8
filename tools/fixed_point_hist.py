#!/usr/bin/env python3
"""Does the fp32 linear-domain Sinkhorn iterate reach a bitwise fixed point within the reference's 100 sweeps?
(numpy emulation on the synthetic third- and fine-level problems)  Prints the share of
problems that do, and the relative change per sweep in float64."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from pats_amd import synth
f32=np.float32
def run(P=256, seed=5, n=65, D=128, third=True):
    inp = synth.third_inputs(seed=seed,P=P) if third else synth.fine_inputs(seed=seed,B=P)
    d0,d1 = inp['d0'],inp['d1']
    D=d0.shape[1]
    S = (np.einsum('bdn,bdm->bnm', d0.astype(np.float64), d1.astype(np.float64))/np.sqrt(D)*0.1).astype(f32)
    ns = inp['scale'] if third else inp['scale_x']*inp['scale_y']
    ns = ns[:,0,:]  # [P,64]
    m = S.shape[1]-1
    ms = f32(m)
    nsum = ns.sum(1)
    norm = -np.log(ms+nsum)
    lognu = np.concatenate([np.log(ns)+norm[:,None], (np.log(ms)+norm)[:,None]],1).astype(f32)
    logmu = np.concatenate([np.repeat(norm[:,None],m,1), (np.log(nsum)+norm)[:,None]],1).astype(f32)
    r = S.max(2,keepdims=True); c=(S-r).max(1,keepdims=True)
    K = np.exp2(((S-r)-c)*f32(1.4426950408889634)).astype(f32)
    mu=np.exp(logmu).astype(f32); nu=np.exp(lognu).astype(f32)
    b=np.exp(c[:,0,:]).astype(f32); a=np.zeros_like(mu)
    conv=np.full(P,-1); per2=np.full(P,-1)
    a_prev2=None;b_prev2=None
    for it in range(100):
        a_new = (mu * (f32(1)/ np.einsum('pij,pj->pi',K,b).astype(f32))).astype(f32)
        b_new = (nu * (f32(1)/ np.einsum('pij,pi->pj',K,a_new).astype(f32))).astype(f32)
        same = (a_new==a).all(1)&(b_new==b).all(1)
        conv[(conv<0)&same]=it
        if a_prev2 is not None:
            s2=(a_new==a_prev2).all(1)&(b_new==b_prev2).all(1)
            per2[(per2<0)&s2]=it
        a_prev2,b_prev2=a,b
        a,b=a_new,b_new
    return conv,per2
for third in (True,False):
    conv,per2=run(P=512 if third else 32, third=third)
    print('third' if third else 'fine')
    print(' fixed-point reached:',(conv>=0).mean(), 'median',np.median(conv[conv>=0]) if (conv>=0).any() else None, 'pcts',np.percentile(conv[conv>=0],[10,50,90,99]) if (conv>=0).any() else None)
    print(' period<=2 reached:',(per2>=0).mean(), np.percentile(per2[per2>=0],[10,50,90,99]) if (per2>=0).any() else None)
def rate(P=256, seed=5, third=True):
    inp = synth.third_inputs(seed=seed,P=P) if third else synth.fine_inputs(seed=seed,B=P)
    d0,d1 = inp['d0'],inp['d1']
    D=d0.shape[1]
    S = (np.einsum('bdn,bdm->bnm', d0.astype(np.float64), d1.astype(np.float64))/np.sqrt(D)*0.1)
    ns = inp['scale'] if third else inp['scale_x']*inp['scale_y']
    ns = ns[:,0,:].astype(np.float64)
    m = S.shape[1]-1
    nsum = ns.sum(1); norm=-np.log(m+nsum)
    nu = np.concatenate([ns*np.exp(norm)[:,None], (m*np.exp(norm))[:,None]],1)
    mu = np.concatenate([np.repeat(np.exp(norm)[:,None],m,1), (nsum*np.exp(norm))[:,None]],1)
    K=np.exp(S - S.max((1,2),keepdims=True))
    b=np.ones_like(nu); a=np.ones_like(mu)
    for it in range(100):
        a_new = mu/np.einsum('pij,pj->pi',K,b)
        b_new = nu/np.einsum('pij,pi->pj',K,a_new)
        if it in (9,19,29,49,74,99):
            rel=np.abs(b_new/b-1).max(1)
            print(it+1, 'rel change per sweep: median %.2e p90 %.2e max %.2e'%(np.median(rel),np.percentile(rel,90),rel.max()))
        a,b=a_new,b_new
print('rate third'); rate(256,third=True)
print('rate fine'); rate(16,third=False)

"""Lane-by-lane NumPy emulation of the index arithmetic of conv_fwd_thin_kernel and conv_wgrad_thin_kernel (csrc/conv_gemm.hip)
against a direct convolution / weight gradient -- the check that was run before the kernels first saw a GPU.
It mirrors the kernels statement by statement (patch staging, row bases, the k pairing, the MFMA operand / result lane
layout, the cross-wave reduction); it does not execute them.  Usage: python tools/emu_thin_kernels.py   (pure Python, minutes)"""

# ---- forward -------------------------------------------------------------------------------------------------------
import numpy as np
def emu_fwd(KH,KW,CT,S,C1,H,W,pad,N=1,seed=0):
    rng=np.random.RandomState(seed)
    Ho=(H+2*pad-KH)//S+1; Wo=(W+2*pad-KW)//S+1
    x=rng.randn(N,H,W,CT).astype(np.float64)      # NHWC, channels = source1 ++ source2
    x1=x[...,:C1].copy(); x2=x[...,C1:].copy(); C2=CT-C1
    w=rng.randn(64,KH,KW,CT)                       # [Cout][KH][KW][Cin]
    bias=rng.randn(64)
    # reference
    xp=np.zeros((N,H+2*pad+S*16+KH,W+2*pad+S*16+KW,CT)); xp[:,pad:pad+H,pad:pad+W]=x
    ref=np.zeros((N,Ho,Wo,64))
    for oy in range(Ho):
        for ox in range(Wo):
            patch=xp[:,oy*S:oy*S+KH,ox*S:ox*S+KW,:]
            ref[:,oy,ox,:]=np.einsum('nhwc,ohwc->no',patch,w)+bias
    # kernel constants
    TH=TW=16; NT=512; BN=64
    K=KH*KW*CT; RL=KW*CT; JP=(RL+1)//2; KP=KH*JP; KS=K|1
    PH=(TH-1)*S+KH; PW=(TW-1)*S+KW; PN=PH*PW*CT
    wflat=w.reshape(64,K)
    wl=np.full(BN*KS+1,np.nan); 
    for idx in range(BN*K):
        col=idx//K; k=idx-col*K; wl[col*KS+k]=wflat.reshape(-1)[idx]
    wl[BN*KS]=0
    y=np.full((N,Ho,Wo,64),np.nan)
    tiles_x=(Wo+TW-1)//TW; tiles_y=(Ho+TH-1)//TH; tiles_img=tiles_x*tiles_y
    for t in range(N*tiles_img):
        n=t//tiles_img; tr=t%tiles_img
        oy0=(tr//tiles_x)*TH; ox0=(tr%tiles_x)*TW
        iy0=oy0*S-pad; ix0=ox0*S-pad
        patch=np.full(PN+4,np.nan); patch[PN:]=0
        for e in range(PN):
            pix=e//CT; c=e-pix*CT; py=pix//PW; px=pix-py*PW
            iy=iy0+py; ix=ix0+px; v=0.0
            if 0<=iy<H and 0<=ix<W:
                v=x1[n,iy,ix,c] if c<C1 else x2[n,iy,ix,c-C1]
            patch[e]=v
        for wid in range(8):
            wm=wid>>1; wn=wid&1
            acc=np.zeros((2,32,32))      # [i][row][col]
            for lane in range(64):
                l31=lane&31; lh=lane>>5
                breg=np.zeros(KP)
                for kp in range(KP):
                    j=2*(kp%JP)+lh
                    v=wl[(wn*32+l31)*KS+(kp//JP)*RL+j]
                    breg[kp]=v if j<RL else 0.0
                for i in range(2):
                    r=wm*64+i*32+l31
                    rb=(((r>>4)*S)*PW+(r&15)*S)*CT+lh
                    for kp in range(KP):
                        off=(kp//JP)*(PW*CT)+2*(kp%JP)
                        a=patch[rb+off]
                        # MFMA: lane (l31,lh) supplies A[row=l31][k=lh] and B[k=lh][col=l31]; store for the product below
                        if lane<32: pass
                    # accumulate emulation done below
            # emulate product properly: D[row][col] += sum_h A[row][h]*B[h][col] per kp
            for i in range(2):
                for kp in range(KP):
                    A=np.zeros((32,2)); B=np.zeros((2,32))
                    for lane in range(64):
                        l31=lane&31; lh=lane>>5
                        r=wm*64+i*32+l31
                        rb=(((r>>4)*S)*PW+(r&15)*S)*CT+lh
                        off=(kp//JP)*(PW*CT)+2*(kp%JP)
                        A[l31,lh]=patch[rb+off]
                        j=2*(kp%JP)+lh
                        v=wl[(wn*32+l31)*KS+(kp//JP)*RL+j]
                        B[lh,l31]=v if j<RL else 0.0
                    acc[i]+=A@B
            for i in range(2):
                for row32 in range(32):
                    row=wm*64+i*32+row32
                    oy=oy0+(row>>4); ox=ox0+(row&15)
                    if oy<Ho and ox<Wo:
                        y[n,oy,ox,wn*32:(wn+1)*32]=acc[i][row32]+bias[wn*32:(wn+1)*32]
    assert not np.isnan(y).any()
    err=np.abs(y-ref).max()/np.abs(ref).max()
    return err
for args in [(7,7,3,1,3,20,18,3),(4,4,3,2,3,36,20,1),(3,3,6,1,3,20,17,1),(3,3,3,1,3,16,16,1),(1,1,12,1,12,17,16,0)]:
    print("fwd", args, emu_fwd(*args))

# ---- weight gradient -----------------------------------------------------------------------------------------------
import numpy as np
def emu_wgrad(KH,KW,CT,S,C1,H,W,pad,N=2,G=3,seed=0):
    rng=np.random.RandomState(seed)
    Ho=(H+2*pad-KH)//S+1; Wo=(W+2*pad-KW)//S+1
    x=rng.randn(N,H,W,CT); dz=rng.randn(N,Ho,Wo,64)
    x1=x[...,:C1]; x2=x[...,C1:]
    K=KH*KW*CT
    # reference
    xp=np.zeros((N,H+2*pad+S*16+KH,W+2*pad+S*16+KW,CT)); xp[:,pad:pad+H,pad:pad+W]=x
    ref=np.zeros((64,KH,KW,CT)); 
    for oy in range(Ho):
        for ox in range(Wo):
            patch=xp[:,oy*S:oy*S+KH,ox*S:ox*S+KW,:]
            ref+=np.einsum('no,nhwc->ohwc',dz[:,oy,ox,:],patch)
    refb=dz.sum((0,1,2))
    TH=TW=16; NT=512; CO=64; NKT=(K+31)//32
    PH=(TH-1)*S+KH; PW=(TW-1)*S+KW; PN=PH*PW*CT
    tiles_x=(Wo+TW-1)//TW; tiles_y=(Ho+TH-1)//TH; tiles_img=tiles_x*tiles_y; ntiles=N*tiles_img
    G=min(G,ntiles)
    total=np.zeros(64*K+64)
    for bx in range(G):
        acc=np.zeros((8,NKT,32,32)); bsum=np.zeros((8,64))
        for t in range(bx,ntiles,G):
            n=t//tiles_img; tr=t%tiles_img
            oy0=(tr//tiles_x)*TH; ox0=(tr%tiles_x)*TW; iy0=oy0*S-pad; ix0=ox0*S-pad
            patch=np.full(PN+4,np.nan)
            for e in range(PN):
                pix=e//CT; c=e-pix*CT; py=pix//PW; px=pix-py*PW; iy=iy0+py; ix=ix0+px; v=0.0
                if 0<=iy<H and 0<=ix<W: v=x1[n,iy,ix,c] if c<C1 else x2[n,iy,ix,c-C1]
                patch[e]=v
            dzl=np.full(TH*TW*CO,np.nan)
            for f in range(TH*TW*CO//4):
                pix=f>>4; c4=f&15; oy=oy0+(pix>>4); ox=ox0+(pix&15)
                v=np.zeros(4)
                if oy<Ho and ox<Wo: v=dz[n,oy,ox,c4*4:c4*4+4]
                dzl[f*4:f*4+4]=v
            for wid in range(8):
                cot=wid&1; sl=wid>>1
                for q in range(32):
                    A=np.zeros((32,2)); B=np.zeros((NKT,2,32))
                    for lane in range(64):
                        l31=lane&31; lh=lane>>5
                        abase=(64*sl+lh)*CO+cot*32+l31
                        a=dzl[abase+2*q*CO]; A[l31,lh]=a; bsum[wid,lane]+=a
                        pbase=((4*sl*S)*PW)*CT+lh*S*CT
                        pq=((q>>3)*S*PW+2*(q&7)*S)*CT
                        for kt in range(NKT):
                            k=kt*32+l31; kk=k if k<K else 0
                            tap=kk//CT; c=kk-tap*CT
                            koff=((tap//KW)*PW+(tap%KW))*CT+c
                            B[kt,lh,l31]=patch[pbase+pq+koff]
                    for kt in range(NKT): acc[wid,kt]+=A@B[kt]
        # reduce
        red=np.full(2*NKT*1024+64,np.nan)
        for rnd in range(4):
            for wid in range(8):
                cot=wid&1; sl=wid>>1
                if sl!=rnd: continue
                for lane in range(64):
                    l31=lane&31; lh=lane>>5
                    bs=bsum[wid,lane]+bsum[wid,lane^32]
                    for kt in range(NKT):
                        for r in range(16):
                            row=(r&3)+8*(r>>2)+4*lh; col=l31
                            idx=((cot*NKT+kt)*16+r)*64+lane
                            red[idx]=(red[idx] if rnd else 0.0)+acc[wid,kt,row,col]
                    if lh==0:
                        idx=2*NKT*1024+cot*32+l31
                        red[idx]=(red[idx] if rnd else 0.0)+bs
        dst=np.zeros(64*K+64)
        for e in range(64*K):
            co=e//K; k=e-co*K; row=co&31; col=k&31
            dst[e]=red[(((co>>5)*NKT+(k>>5))*16+(row&3)+4*(row>>3))*64+col+32*((row>>2)&1)]
        dst[64*K:]=red[2*NKT*1024:2*NKT*1024+64]
        total+=dst
    assert not np.isnan(total).any()
    e1=np.abs(total[:64*K].reshape(64,KH,KW,CT)-ref).max()/np.abs(ref).max()
    e2=np.abs(total[64*K:]-refb).max()/np.abs(refb).max()
    return e1,e2
for args in [(7,7,3,1,3,20,18,3),(4,4,3,2,3,36,20,1),(3,3,6,1,3,20,17,1),(3,3,3,1,3,16,16,1),(1,1,12,1,12,17,16,0)]:
    print("wgrad", args, emu_wgrad(*args))

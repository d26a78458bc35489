import numpy as np, torch, ctypes, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import mmdet3d_gaussian_amd as amd, oracle
from rbox_inputs import nms_boxes
lib=amd.load_library()
n,thr=int(sys.argv[1]) if len(sys.argv)>1 else 1000,0.25
boxes,scores=nms_boxes(n,seed=77)
order=np.argsort(-scores,kind='stable'); bs=np.ascontiguousarray(boxes[order])
want=oracle.nms_mask(bs,thr)
d=torch.from_numpy(bs).cuda()
keep=torch.empty(n,dtype=torch.int64,device='cuda'); num=torch.zeros(1,dtype=torch.int64,device='cuda')
ws=torch.zeros(lib.rnms_workspace_bytes(n),dtype=torch.uint8,device='cuda')
vp=lambda t: ctypes.c_void_p(t.data_ptr())
print('rc',lib.rnms_bev(vp(d),n,thr,vp(keep),vp(num),vp(ws),None)); torch.cuda.synchronize()
cb=(n+63)//64; off=(n*64+255)//256*256
ob=ws[:n*64].cpu().numpy().view(np.float32).reshape(n,16)
mask=ws[off:off+n*cb*8].cpu().numpy().view(np.uint64).reshape(n,cb)
rows=np.arange(n)[:,None]//64; cols=np.arange(cb)[None,:]; upper=cols>=rows
diff=(mask!=want)&upper
print('mask mismatching words',diff.sum(),'of',upper.sum())
# compare obox with numpy recompute of sincos
iou_o=oracle.iou_bev_xyxyr(bs[:200],bs[:200])
iou_g=amd.boxes_iou_bev(d[:200],d[:200]).cpu().numpy()
print('iou max abs diff',np.abs(iou_o-iou_g).max(),'n diff',(iou_o!=iou_g).sum())
k=int(num.item()); wk=oracle.nms_bev(bs,thr)
print('keep count gpu',k,'oracle',len(wk),'equal',np.array_equal(keep[:k].cpu().numpy(),wk))
if diff.sum():
    i,c=np.argwhere(diff)[0]; print('first diff row',i,'col',c,hex(int(mask[i,c])),hex(int(want[i,c])))

"""SSD300 / VGG16-bn topology as data (what the reference builds imperatively in model/vgg16.py:25-37 and
model/ssd.py:28-77, 118).  Shared by the nn.Module shell (state_dict layout) and the HIP engine (launch plan)."""

# VGG16-bn `features` cfg (vgg16.py:25): ints = conv3x3(p1)+BN+ReLU, 'M' = MaxPool(2,2), 'M_P' = MaxPool(2,2,padding=1)
VGG_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M_P', 512, 512, 512, 'M', 512, 512, 512, 'M_P')
# extras (ssd.py:49-54): per stage a list of (kernel, cout, stride, pad)
EXTRA_CFG = (((3, 1024, 1, 1),), ((1, 1024, 1, 0),), ((1, 256, 1, 0), (3, 512, 2, 1)), ((1, 128, 1, 0), (3, 256, 2, 1)),
             ((1, 128, 1, 0), (3, 256, 1, 0)), ((1, 128, 1, 0), (3, 256, 1, 0)))
# detectors (ssd.py:70-77): (source activation tag, cin, anchors per cell)
HEAD_CFG = (('4_3', 512, 4), ('7_1', 1024, 6), ('8_2', 512, 6), ('9_2', 256, 6), ('10_2', 256, 4), ('11_2', 256, 4))
# default-box grid (ssd.py:118) and scales (ssd.py:114)
BOX_MAPS = ((38, 38, 4), (19, 19, 6), (10, 10, 6), (5, 5, 6), (3, 3, 4), (1, 1, 4))
S_MIN, S_MAX = 0.2, 0.9
MATCH_THRESHOLD = 0.25      # ssd.py:231
NMS_IOU_THRESHOLD = 0.5     # utils.py:80


class Topology:
    """One input-size variant.  SSD300 is the reference's network; SSD512 is a build-side extension (the reference hard-codes
    the 300-px grid, ssd.py:118): BASELINE configs[4]'s 24,564-prior layout = maps 64/32/16/8/4/2/1 with 4/6/6/6/6/4/4
    anchors, which needs pool_3 without padding on the even 128-px map and five stride-2 extras (one extra stage)."""

    def __init__(self, imsize, vgg_cfg, extra_cfg, head_cfg, box_maps, s_min=S_MIN, s_max=S_MAX, keep_pool5=False, vgg_frozen=True):
        self.imsize, self.vgg_cfg, self.extra_cfg, self.head_cfg, self.box_maps = imsize, vgg_cfg, extra_cfg, head_cfg, box_maps
        self.s_min, self.s_max = s_min, s_max
        self.keep_pool5, self.vgg_frozen = keep_pool5, vgg_frozen

    def num_priors(self):
        return sum(m * n * a for m, n, a in self.box_maps)


SSD300 = Topology(300, VGG_CFG, EXTRA_CFG, HEAD_CFG, BOX_MAPS)
_S2 = ((1, 128, 1, 0), (3, 256, 2, 1))
SSD512 = Topology(
    512,
    (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M_P'),
    (((3, 1024, 1, 1),), ((1, 1024, 1, 0),), ((1, 256, 1, 0), (3, 512, 2, 1)), _S2, _S2, _S2, _S2),
    (('4_3', 512, 4), ('7_1', 1024, 6), ('8_2', 512, 6), ('9_2', 256, 6), ('10_2', 256, 6), ('11_2', 256, 4), ('12_2', 256, 4)),
    ((64, 64, 4), (32, 32, 6), (16, 16, 6), (8, 8, 6), (4, 4, 6), (2, 2, 4), (1, 1, 4)))


# the VGG16-bn classifier's feature stack (model/vgg16.py:22-39): every layer of the cfg incl. the last pool, nothing frozen, no
# extras / heads / default boxes; any input size (the classifier needs a 7x7 final map: 184..215 px with these pools)
VGG16_FEATURES = Topology(None, VGG_CFG, (), (), (), keep_pool5=True, vgg_frozen=False)
VGG16_FEATURES_FROZEN = Topology(None, VGG_CFG, (), (), (), keep_pool5=True, vgg_frozen=True)      # transfer_learning=True (vgg16.py:72-75)


def topology_for(imsize):
    if imsize == 300:
        return SSD300
    if imsize == 512:
        return SSD512
    raise ValueError(f'imsize must be 300 (reference) or 512 (extension), got {imsize}')


def feature_plan(topo=SSD300):
    """Ordered list of feature layers.  conv: name,cin,cout,k,stride,pad,frozen; bn: name,c,frozen; act; pool: name,pad.
    SSD: pool_5 is dropped and every VGG layer is frozen (ssd.py:31-41); the VGG16 classifier keeps pool_5 and trains everything."""
    plan, cin, stage, sub = [], 3, 1, 1
    for v in topo.vgg_cfg:
        if isinstance(v, str):
            if stage < 5 or topo.keep_pool5:
                plan.append(dict(kind='pool', name=f'pool_{stage}', pad=1 if v == 'M_P' else 0))
            stage, sub = stage + 1, 1
            continue
        tag = f'{stage}_{sub}'
        plan.append(dict(kind='conv', name=f'conv_{tag}', tag=tag, cin=cin, cout=v, k=3, stride=1, pad=1, frozen=topo.vgg_frozen))
        plan.append(dict(kind='bn', name=f'bn_{tag}', tag=tag, c=v, frozen=topo.vgg_frozen))
        plan.append(dict(kind='act', name=f'act_{tag}', tag=tag))
        cin, sub = v, sub + 1
    for group in topo.extra_cfg:
        sub = 1
        for (k, v, s, p) in group:
            tag = f'{stage}_{sub}'
            plan.append(dict(kind='conv', name=f'conv_{tag}', tag=tag, cin=cin, cout=v, k=k, stride=s, pad=p, frozen=False))
            plan.append(dict(kind='bn', name=f'bn_{tag}', tag=tag, c=v, frozen=False))
            plan.append(dict(kind='act', name=f'act_{tag}', tag=tag))
            cin, sub = v, sub + 1
        stage += 1
    return plan


def head_plan(num_classes, topo=SSD300):
    return [dict(name=f'det_{t}', tag=t, cin=c, anchors=a, cout=a * (num_classes + 4)) for t, c, a in topo.head_cfg]


def num_priors(topo=SSD300):
    return topo.num_priors()

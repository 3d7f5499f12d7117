#!/usr/bin/env python3
"""Author the yolov3-tiny INT8 cfg files this repo ships (our own text; same cfg *keys* as the reference parser
accepts: /root/reference/src/parser.c:170-204 (convolutional), 411-431 (maxpool), 506-518 (upsample),
520-564 (route), 254-291 (yolo), 579-674 (net)).

  cfg/yolov3-tiny_quant.cfg        leaky activations (the variant BASELINE.json names)
  cfg/yolov3-tiny_quant_relu6.cfg  relu6 activations (same topology as the one cfg the reference ships)
  cfg/tiny_unit.cfg                12x12 5-layer unit-test net (conv3x3, maxpool 2/2, conv1x1, maxpool 2/1, ...)
  cfg/yolov3_chain_quant.cfg       608x608 chain of full YOLOv3's conv shapes (5 stride-2 stages, 1x1/3x3 pairs, 255-channel head)
  cfg/yolov3_quant.cfg             full YOLOv3 @608 (75 convs, 23 quantized [shortcut]s, 3 heads) -- BASELINE config[4]
  cfg/res_unit.cfg                 16x16 unit net for the quantized residual add and the glue layers' quant_stop tails
  cfg/s2_unit.cfg                  24x24 chain of stride-2 3x3 convolutions (the downsampling layers of full YOLOv3)
"""
import os, sys

def net(w, h, c=3, batch=1):
    return f"[net]\nbatch={batch}\nsubdivisions=1\nwidth={w}\nheight={h}\nchannels={c}\n\n"

def conv(filters, size, act, bn=1, stop=0, stride=1):
    s = "[convolutional]\n"
    if bn: s += "batch_normalize=1\n"
    s += f"filters={filters}\nsize={size}\nstride={stride}\npad=1\nactivation={act}\nquantized=1\nquant_stop={stop}\n\n"
    return s

def maxpool(size, stride):
    return f"[maxpool]\nsize={size}\nstride={stride}\nquantized=1\nquant_stop=0\n\n"

def route(layers):
    return f"[route]\nlayers = {layers}\nquantized=1\nquant_stop=0\n\n"

def upsample(stride=2):
    return f"[upsample]\nstride={stride}\nquantized=1\nquant_stop=0\n\n"

def yolo(mask, classes=5):
    return (f"[yolo]\nmask = {mask}\nanchors = 10,14,  23,27,  37,58,  81,82,  135,169,  344,319\n"
            f"classes={classes}\nnum=6\njitter=.3\nignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n\n")

def yolov3_tiny(act, classes=5, w=416, h=416):
    nout = 3 * (classes + 5)
    s = net(w, h)
    for f in (16, 32, 64, 128, 256):
        s += conv(f, 3, act) + maxpool(2, 2)
    s += conv(512, 3, act) + maxpool(2, 1)
    s += conv(1024, 3, act) + conv(256, 1, act) + conv(512, 3, act)
    s += conv(nout, 1, "linear", bn=0, stop=1) + yolo("3,4,5", classes)
    s += route("-4") + conv(128, 1, act) + upsample(2) + route("-1, 8")
    s += conv(256, 3, act) + conv(nout, 1, "linear", bn=0, stop=1) + yolo("0,1,2", classes)
    return s

def tiny_unit(act="leaky"):
    # 12x12x3 input, exercises the first-layer kernel, every quantised layer type, both maxpool geometries and a
    # multi-input route
    s = net(12, 12, c=3)
    s += conv(16, 3, act)            # 0
    s += maxpool(2, 2)               # 1   6x6
    s += conv(32, 3, act)            # 2
    s += maxpool(2, 1)               # 3   6x6 (pad 1, offset 0)
    s += conv(16, 1, act)            # 4
    s += upsample(2)                 # 5   12x12
    s += route("-1, 0")              # 6   32ch 12x12
    s += conv(32, 3, "relu6")        # 7
    s += conv(30, 1, "linear", bn=0, stop=1)  # 8
    s += yolo("0,1,2")               # 9
    return s

def s2_unit(act="leaky"):
    # 24x24x3 input, a darknet-53-style downsampling chain: stride-2 3x3 convolutions instead of maxpools (the layer
    # shapes of full YOLOv3, BASELINE config[4], in miniature), odd map at the end
    s = net(24, 24, c=3)
    s += conv(16, 3, act)                 # 0   24x24 (first-layer kernel)
    s += conv(32, 3, act, stride=2)       # 1   12x12
    s += conv(16, 1, act)                 # 2
    s += conv(64, 3, act, stride=2)       # 3   6x6   (16-byte channel chunks)
    s += conv(128, 3, "relu6", stride=2)  # 4   3x3   (64-byte channel chunks)
    s += conv(64, 1, act)                 # 5
    s += conv(128, 3, act, stride=2)      # 6   2x2   (odd input map)
    s += conv(30, 1, "linear", bn=0, stop=1)  # 7
    s += yolo("0,1,2")                    # 8
    return s

def shortcut(frm=-3):
    return f"[shortcut]\nfrom={frm}\nactivation=linear\nquantized=1\nquant_stop=0\n\n"

YOLOV3_ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"

def yolov3(act="leaky", classes=80, w=608, h=608):
    # Full YOLOv3 (darknet-53 trunk + 3 detection heads, BASELINE config[4]): 75 convolutions, 23 residual [shortcut]s,
    # 4 routes, 2 upsamples, 3 yolo layers = 107 layers, the public yolov3.cfg topology with the fork's quant keys on
    # every layer.  `[shortcut] quantized=1` is this build's own integer op (the reference's shortcut is float only).
    def yolo3(mask):
        return (f"[yolo]\nmask = {mask}\nanchors = {YOLOV3_ANCHORS}\nclasses={classes}\nnum=9\njitter=.3\n"
                f"ignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n\n")
    nout = 3 * (classes + 5)
    s = net(w, h)
    s += conv(32, 3, act)
    for f, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        s += conv(f, 3, act, stride=2)
        for _ in range(blocks):
            s += conv(f // 2, 1, act) + conv(f, 3, act) + shortcut(-3)
    for f, mask, lat in ((1024, "6,7,8", 61), (512, "3,4,5", 36), (256, "0,1,2", None)):
        for _ in range(3):
            s += conv(f // 2, 1, act) + conv(f, 3, act)
        s += conv(nout, 1, "linear", bn=0, stop=1) + yolo3(mask)
        if lat is not None:
            s += route("-4") + conv(f // 4, 1, act) + upsample(2) + route(f"-1, {lat}")
    return s

def res_unit(act="leaky"):
    # 16x16 unit net for the quantized residual add: two residual blocks (one on 32, one on 64 channels), a route that
    # concatenates a shortcut's output (as YOLOv3's `route -1, 61` does), quant_stop tails on a maxpool and a 2-input route
    s = net(16, 16, c=3)
    s += conv(32, 3, act)                 # 0
    s += conv(16, 1, act)                 # 1
    s += conv(32, 3, act)                 # 2
    s += shortcut(-3)                     # 3   = L2 + L0
    s += conv(64, 3, act, stride=2)       # 4   8x8
    s += conv(32, 1, act)                 # 5
    s += conv(64, 3, "relu6")             # 6
    s += shortcut(-3)                     # 7   = L6 + L4 (different scales / zero points)
    s += conv(32, 1, act)                 # 8
    s += upsample(2)                      # 9   16x16
    s += route("-1, 3")                   # 10  64ch
    s += "[maxpool]\nsize=2\nstride=2\nquantized=1\nquant_stop=1\n\n"   # 11  8x8, dequant tail
    s += "[route]\nlayers = -1, 7\nquantized=1\nquant_stop=1\n\n"         # 12  128ch, dequant tail per input
    s += conv(30, 1, "linear", bn=0, stop=1)  # 13
    s += yolo("0,1,2")                    # 14
    return s

def yolov3_chain(act="leaky", classes=80, w=608, h=608):
    # the convolution shapes of full YOLOv3's darknet-53 trunk at 608x608 (BASELINE config[4]) as a plain chain: five
    # stride-2 3x3 downsampling convs, the 1x1 / 3x3 pair of every residual stage once, one detection head.  The
    # reference has no quantized [shortcut] (src/shortcut_layer.c is float only), so the residual adds are left out.
    s = net(w, h)
    s += conv(32, 3, act)
    for f in (64, 128, 256, 512, 1024):
        s += conv(f, 3, act, stride=2) + conv(f // 2, 1, act) + conv(f, 3, act)
    s += conv(3 * (classes + 5), 1, "linear", bn=0, stop=1)
    s += (f"[yolo]\nmask = 6,7,8\nanchors = 10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326\n"
          f"classes={classes}\nnum=9\njitter=.3\nignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n\n")
    return s

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cfg")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "yolov3-tiny_quant.cfg"), "w").write(yolov3_tiny("leaky"))
    open(os.path.join(out, "yolov3-tiny_quant_relu6.cfg"), "w").write(yolov3_tiny("relu6"))
    open(os.path.join(out, "tiny_unit.cfg"), "w").write(tiny_unit())
    open(os.path.join(out, "s2_unit.cfg"), "w").write(s2_unit())
    open(os.path.join(out, "yolov3_chain_quant.cfg"), "w").write(yolov3_chain())
    open(os.path.join(out, "yolov3_quant.cfg"), "w").write(yolov3())
    open(os.path.join(out, "res_unit.cfg"), "w").write(res_unit())
    print("wrote cfgs to", os.path.normpath(out))

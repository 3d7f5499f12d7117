#!/usr/bin/env python3
"""Author the yolov3-tiny INT8 cfg files this repo ships (our own text; same cfg *keys* as the reference parser
accepts: /root/reference/src/parser.c:170-204 (convolutional), 411-431 (maxpool), 506-518 (upsample),
520-564 (route), 254-291 (yolo), 579-674 (net)).

  cfg/yolov3-tiny_quant.cfg        leaky activations (the variant BASELINE.json names)
  cfg/yolov3-tiny_quant_relu6.cfg  relu6 activations (same topology as the one cfg the reference ships)
  cfg/tiny_unit.cfg                12x12 5-layer unit-test net (conv3x3, maxpool 2/2, conv1x1, maxpool 2/1, ...)
  cfg/yolov3_chain_quant.cfg       608x608 chain of full YOLOv3's conv shapes (5 stride-2 stages, 1x1/3x3 pairs, 255-channel head)
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
    print("wrote cfgs to", os.path.normpath(out))

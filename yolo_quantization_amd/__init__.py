"""MI355X-native INT8 quantized-convolution inference path for the darknet uint8-quantization fork
ArtyZe/yolo_quantization.  See DESIGN.md; the product is the HIP C-ABI library (csrc/) and the plain-C host
(host/); this Python package only carries tooling (synthetic models, ctypes bindings for tests / bench)."""

"""MI355X-native counterpart of the reference's fake-quant package `qdiff` (ViDiT-Q/quant_utils/qdiff): the same
class names, attributes and model-surgery helpers, but a quantized layer's forward really runs int8
(per-token quantize kernel -> int8-MFMA GEMM with the dequant epilogue) instead of simulating it with
quantize->dequantize->F.linear.  The two are the same function up to fp32 rounding:
F.linear(q_a*da, (q_w+zp)*dw) + b == (q_a.q_w)*da*dw + da*sum(q_a)*zp*dw + b."""

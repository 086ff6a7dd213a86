from ._placeholder import placeholder

AdaGroupNorm = placeholder("AdaGroupNorm")
AdaLayerNorm = placeholder("AdaLayerNorm")
AdaLayerNormContinuous = placeholder("AdaLayerNormContinuous")
AdaLayerNormZero = placeholder("AdaLayerNormZero")
AdaLayerNormSingle = placeholder("AdaLayerNormSingle")
RMSNorm = placeholder("RMSNorm")

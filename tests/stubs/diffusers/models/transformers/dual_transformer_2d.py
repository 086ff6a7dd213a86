from .._placeholder import placeholder

DualTransformer2DModel = placeholder("DualTransformer2DModel")

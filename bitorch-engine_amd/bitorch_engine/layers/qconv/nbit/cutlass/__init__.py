from .layer import Q4Conv2dCutlass

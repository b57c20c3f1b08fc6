"""Build-container stub so modules that `import cv2` at top level can be imported; no function is provided."""
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_AREA = 3
INTER_CUBIC = 2

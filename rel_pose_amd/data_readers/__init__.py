"""Dataset side of the path: readers with the reference's class names, constructor arguments,
metadata file layout and sample convention -- `(images [2,3,H,W] fp32 BGR 0..255, poses [2,7] (t, q xyzw), intrinsics [2,4])`
(reference src/data_readers/base.py:45-97) -- and a colour/resize augmentor that runs on whatever device its input is on."""

"""The one piece of the reference's XLFMDataset.py that sits directly in front of the hot path:
cropping the 29 lenslet views out of the sensor frame.  Not registered by cwfa_amd.install() (the reference's own module
holds the dataset classes); a maintainer patches the static method, see INTEGRATION.md."""
from . import ops

__all__ = ["XLFMDatasetFull", "extract_views"]


def extract_views(image, lenslet_coords, subimage_shape, debug=False):
    """XLFMDatasetFull.extract_views, XLFMDataset.py:212-242 (same signature; ``debug`` draws markers in the reference and
    is not supported here)."""
    if debug:
        raise NotImplementedError("extract_views(debug=True) is a plotting aid of the reference")
    return ops.extract_views(image, lenslet_coords, subimage_shape)


class XLFMDatasetFull:
    extract_views = staticmethod(extract_views)

    @staticmethod
    def extract_views_normalized(image, lenslet_coords, subimage_shape, mean_imgs, std_imgs):
        """extract_views followed by ``(views - mean_imgs) / std_imgs`` (CWFA.py:796-797), fused."""
        return ops.extract_views(image, lenslet_coords, subimage_shape, float(mean_imgs), float(std_imgs))

"""`python -m rendernet_amd.tools.create_TAR --images_path DIR --save_path out.tar` -- the reference's data-set packer
(tools/create_TAR.py): every file of `images_path` matching `file_format` goes into one tar under its base name, which is
the container `NpyTarReader` / `data_loader` stream from.  (The reference reads `args.imgages_path` [sic], :36, and so
fails before adding anything; this one uses the argument it declares.)"""
import argparse
import glob
import os
import tarfile


def create_tar(images_path, save_path, file_format='*.png', to_compress=False):
    """Returns the number of members written."""
    files = sorted(glob.glob(os.path.join(images_path, file_format)))
    with tarfile.open(save_path, "w:gz" if to_compress else "w") as tar:
        for item in files:
            tar.add(item, arcname=os.path.basename(item), recursive=False)
    return len(files)


def main(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--images_path', type=str, required=True, help="Path to the image directory.")
    parser.add_argument('--save_path', type=str, required=True, help="Path to save TAR file")
    parser.add_argument('--file_format', type=str, default='*.png', help="Image format")
    parser.add_argument('--to_compress', action='store_true', help="Compress .tar file or not, useful to move files around")
    args = parser.parse_args(argv)
    n = create_tar(args.images_path, args.save_path, args.file_format, args.to_compress)
    print("Found {0} images".format(n))


if __name__ == "__main__":
    main()

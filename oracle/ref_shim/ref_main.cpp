/* ref_main.cpp — entry point of oracle/_ref/bam-readcount-ref: the reference's own main() (bamreadcount.cpp:421-670,
 * compiled unmodified inside ref_driver.cpp under the name brc_reference_main) on top of the shim.  Test infrastructure. */
int brc_reference_main(int argc, char* argv[]);
int main(int argc, char* argv[]) { return brc_reference_main(argc, argv); }
